#!/bin/bash
# Developer tool (this container, after `gpurun -- bash tools/r6_measure.sh r6_final full`): copies the summaries of the
# measurement set from gpurun_out/ (scratch) into profiles/ (tracked) under the names bench.py and the README use.
set -eu
T=${1:-r6_final}
O=gpurun_out/$T
for f in bench.json bench_single_lane.json bench_under_rocprof.json pmc_bench_under_rocprof.json kernel_stats.csv \
         pmc_per_kernel.json pmc_per_kernel.txt pmc_sq_per_kernel.json pmc_sq_per_kernel.txt pmc_occupancy_per_kernel.txt pmc_l2_per_kernel.txt; do
  [ -f $O/${T}_$f ] && cp $O/${T}_$f profiles/${T}_$f
done
cp $O/${T}_pmc_per_kernel.json profiles/r6_pmc_per_kernel.json       # what bench.py reads (roofline.traffic, gcups)
cp $O/${T}_pmc_sq_per_kernel.json profiles/r6_pmc_sq_per_kernel.json # what roofline_top5 reads
if [ -d gpurun_out/r6g ]; then cp gpurun_out/r6g/r6_*.txt profiles/; fi
python3 tools/kernel_resources.py > profiles/r6_kernel_resources.txt 2>/dev/null
python3 - <<'PY'
import json
from lancet2_amd.stamp import csrc_sha16
p = json.load(open("profiles/r6_pmc_per_kernel.json"))
print("profiles/r6_pmc_per_kernel.json stamp", p["_stamp"], "sources now", csrc_sha16(), "FRESH" if p["_stamp"]["csrc_sha16"] == csrc_sha16() else "STALE")
PY
