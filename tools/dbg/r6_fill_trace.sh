#!/bin/bash
# Developer tool (GPU box): do k_poa and its fill workers overlap?  kernel-trace timestamps of one POA-only run.
set -u
R=$PWD
O=$R/gpurun_out/r6_fill_trace
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MA_POA_FILL_WGS=6 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/tools/poa_bench.py 8192 256 > $O/log.txt 2>&1
cd $R
python3 - <<'PY'
import csv, glob
rows = []
for p in glob.glob("gpurun_out/r6_fill_trace/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "k_poa" in r["Kernel_Name"] or "k_msa_maxima" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id"), r.get("Stream_Id")))
rows.sort()
t0 = rows[0][0] if rows else 0
for s, e, n, q, st in rows[-12:]:
    print("%10.3f ms  +%8.3f ms  q=%s s=%s %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, st, n))
PY
