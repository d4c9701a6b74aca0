#!/bin/bash
# Developer tool: POA with band-sized window areas + per-workgroup full-fill areas -- the POA tests, then the bench (4 lanes, 1 lane)
set -u
O=gpurun_out/r6_areas
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_wide_components.py tests/test_gpu_variants.py -x -q -m gpu 2>&1 | tail -3
export MA_BENCH_CACHE=/tmp/ma_bench_cache
python3 bench.py --no-cpu --no-also --gen-only > $O/gen.log 2>&1
MA_VERBOSE=1 timeout 300 python3 bench.py --no-cpu --no-also --gen-workers 1 --steps 1 --warmup 1 2>&1 | grep "msa:" | sort | uniq -c | head -3
for rep in 1 2; do
  for lanes in 4 1; do
    MA_STREAMS=$lanes timeout 300 python3 bench.py --no-cpu --no-also --gen-workers 1 2>/dev/null | tail -1 > $O/b_${lanes}_$rep.json
    python3 - <<P
import json
d=json.load(open("$O/b_${lanes}_$rep.json"))
print("lanes $lanes rep $rep", d["value"], d["ms_per_step"], "k_poa", d["kernel_ms_per_step"].get("k_poa"), d["parity_sample"])
P
  done
done
