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
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            m = re.search(r"(k_[a-z_0-9]+|gate_kernel)", row["Kernel_Name"])
            name = m.group(1) if m else row["Kernel_Name"].split("(")[0][:40]
            if pat and not pat.search(name):
                continue
            tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[name].add(row["Dispatch_Id"])
for name in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0.0)):
    n = max(len(disp[name]), 1)
    print(f"{name:20s} launches {n:4d} " + " ".join(f"{c}={v / n:.4g}" for c, v in sorted(tot[name].items())))
