"""Per-kernel PMC figures from separate rocprofv3 --pmc passes of the SAME bench.py command (FETCH_SIZE, WRITE_SIZE and
SQ_INSTS_VALU do not share a pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots"):

  rocprofv3 --pmc FETCH_SIZE    --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --no-cpu --no-also
  rocprofv3 --pmc WRITE_SIZE    --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --no-cpu --no-also
  rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_valu  -- python3 bench.py --steps 2 --no-cpu --no-also
  python3 tools/pmc_per_kernel.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_valu profiles/r2_pmc_per_kernel.json

Units / corrections (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies
128-byte requests at 64 bytes, so reads are doubled (exact for 16 B/lane streaming reads, an upper estimate for
narrower patterns); WRITE_SIZE is taken as is.  Infinity-Cache hits are counted as traffic.  SQ_INSTS_VALU counts
wave-level vector instructions.  Everything is averaged per launch of each kernel NAME (all template instantiations
and all launches of the run, statistics and warm-up steps included) -- the way bench.py aggregates its HIP-event times.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lancet2_amd.stamp import csrc_sha16  # noqa: E402


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
    valu = collect(sys.argv[3], "SQ_INSTS_VALU")
    out = {}
    for k in sorted(set(fetch) | set(write) | set(valu)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        vi, nv = valu.get(k, (0.0, 0))
        rd = f * 1024.0 * 2.0 / max(nf, 1)
        wr = w * 1024.0 / max(nw, 1)
        out[k] = {"launches": max(nf, nw, nv), "read_bytes_per_launch": int(rd), "write_bytes_per_launch": int(wr),
                  "bytes_per_launch": int(rd + wr), "valu_insts_per_launch": int(vi / max(nv, 1))}
    out["_note"] = ("FETCH_SIZE x 1024 x 2 (gfx950 read correction) + WRITE_SIZE x 1024; SQ_INSTS_VALU wave instructions; "
                    "averaged per launch; three separate --pmc passes of `python3 bench.py --steps 2 --no-cpu --no-also`")
    # provenance: bench.py marks these figures stale when the kernel sources it runs are not the ones profiled here
    # provenance = what the PROFILED run loaded (the `build` object of its own JSON line, argv[5] = that run's log), not what
    # the sources look like when this script runs afterwards
    stamp = {"csrc_sha16": csrc_sha16()}
    if len(sys.argv) > 5 and os.path.exists(sys.argv[5]):
        for line in open(sys.argv[5]):
            if line.startswith('{"metric"'):
                stamp = json.loads(line).get("build", stamp)
    stamp["command"] = "python3 bench.py --steps 2 --no-cpu --no-also --gen-workers 1"
    stamp["steps_profiled"] = 5  # 2 timed + 2 warm-up + 1 statistics step: what `launches` counts
    out["_stamp"] = stamp
    json.dump(out, open(sys.argv[4], "w"), indent=1, sort_keys=True)
    for k, v in out.items():
        if not k.startswith("_"):
            print(f"{k:24s} launches {v['launches']:3d}  read {v['read_bytes_per_launch'] / 1e9:9.3f} GB  "
                  f"write {v['write_bytes_per_launch'] / 1e9:9.3f} GB  valu {v['valu_insts_per_launch'] / 1e6:10.2f} M")


if __name__ == "__main__":
    main()
