#!/bin/bash
# Developer tool: k_poa's start order (MA_POA_ORDER=1 most alignments first / 0 window order) -- parity tests, then A/B
set -u
O=gpurun_out/r6_order
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_wide_components.py -x -q -m gpu -k "msa or poa or wide" 2>&1 | tail -2
export MA_BENCH_CACHE=/tmp/ma_bench_cache
python3 bench.py --no-cpu --no-also --gen-only > $O/gen.log 2>&1
for rep in 1 2; do
  for ord in 1 0; do
    for lanes in 4 1; do
      MA_POA_ORDER=$ord MA_STREAMS=$lanes timeout 300 python3 bench.py --no-cpu --no-also --gen-workers 1 2>/dev/null | tail -1 > $O/b_${ord}_${lanes}_$rep.json
      python3 - <<P
import json
d=json.load(open("$O/b_${ord}_${lanes}_$rep.json"))
print("order $ord lanes $lanes rep $rep", d["value"], d["ms_per_step"], "k_poa", d["kernel_ms_per_step"].get("k_poa"))
P
    done
  done
done
