"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are collected in separate
runs: they do not fit one pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots").

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --no-cpu
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --no-cpu
  python3 tools/hbm_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r1_hbm_traffic.json

Units / corrections (MI355X_MICROARCH.md, "HBM"): both counters are in KiB; on gfx950 FETCH_SIZE tallies
128-byte requests at 64 bytes, so reads are doubled (exact for 16 B/lane streaming reads, uncalibrated -- an
upper estimate -- for narrower patterns); WRITE_SIZE is taken as is.  Infinity-Cache hits are counted as
traffic.  Values are averaged per launch of each kernel (all launches of the run, warm-up included).
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def collect(root, counter):
    tot, disp = defaultdict(float), defaultdict(set)
    for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                m = re.search(r"(k_[a-z_0-9]+|gate_kernel)", row["Kernel_Name"])
                name = m.group(1) if m else row["Kernel_Name"].split("(")[0][:40]
                tot[name] += float(row["Counter_Value"])
                disp[name].add(row["Dispatch_Id"])
    return {k: (tot[k], len(disp[k])) for k in tot}


def main():
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        rd = f * 1024.0 * 2.0 / max(nf, 1)
        wr = w * 1024.0 / max(nw, 1)
        out[k] = {"launches": max(nf, nw), "read_bytes_per_launch": int(rd), "write_bytes_per_launch": int(wr),
                  "bytes_per_launch": int(rd + wr)}
    out["_note"] = ("FETCH_SIZE x 1024 x 2 (gfx950 read correction) + WRITE_SIZE x 1024, averaged per launch; "
                    "collected in two separate --pmc passes of `python3 bench.py --steps 2 --no-cpu`")
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in out.items():
        if k != "_note":
            print(f"{k:24s} launches {v['launches']:3d}  read {v['read_bytes_per_launch'] / 1e9:9.3f} GB  "
                  f"write {v['write_bytes_per_launch'] / 1e9:9.3f} GB")


if __name__ == "__main__":
    main()
