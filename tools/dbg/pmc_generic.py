"""Developer tool: sums of arbitrary rocprofv3 --pmc counters per kernel name, averaged per launch.

  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... --kernel-trace --output-format csv -d gpurun_out/pmc_x -- python3 bench.py ...
  python3 tools/dbg/pmc_generic.py gpurun_out/pmc_x [kernel-regex]
"""
import csv
import glob
import re
import sys
from collections import defaultdict

root = sys.argv[1]
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "-" else None
json_out = sys.argv[3] if len(sys.argv) > 3 else None  # also write {kernel: {counter: mean per launch}}
tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
dur = defaultdict(dict)  # kernel -> dispatch -> ns (when the csv carries the dispatch's timestamps)
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            m = re.search(r"(k_[a-z_0-9]+|gate_kernel)", row["Kernel_Name"])
            name = m.group(1) if m else row["Kernel_Name"].split("(")[0][:40]
            if pat and not pat.search(name):
                continue
            tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[name].add(row["Dispatch_Id"])
            if row.get("Start_Timestamp") and row.get("End_Timestamp"):
                dur[name][row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
for name in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0.0)):
    n = max(len(disp[name]), 1)
    extra = ""
    if dur[name]:
        ns = sum(dur[name].values()) / len(dur[name])
        extra = f" avg_ms={ns / 1e6:.3f}"
        if "SQ_BUSY_CU_CYCLES" in tot[name] and ns > 0:  # mean number of CUs with a wave on them while the kernel runs (2.4 GHz)
            extra += f" busy_CUs={tot[name]['SQ_BUSY_CU_CYCLES'] / n / (ns * 2.4):.1f}"
    print(f"{name:20s} launches {n:4d} " + " ".join(f"{c}={v / n:.4g}" for c, v in sorted(tot[name].items())) + extra)

if json_out:
    import json
    json.dump({name: {c: v / max(len(disp[name]), 1) for c, v in tot[name].items()} for name in tot}, open(json_out, "w"), indent=1, sort_keys=True)
