#!/bin/bash
# Developer tool: the POA knobs again in the two-lane regime
export MA_BENCH_CACHE=/tmp/ma_bench_cache
O=gpurun_out/r6_knobs2
mkdir -p $O
python3 bench.py --no-cpu --no-also --gen-only > $O/gen.log 2>&1
run() {
  timeout 300 python3 bench.py --no-cpu --no-also --gen-workers 1 2>$O/err.txt | tail -1 > $O/b.json
  python3 - <<P
import json
d=json.load(open("$O/b.json"))
k=d["kernel_ms_per_step"]
print("$1", d["value"], d["ms_per_step"], "k_poa", k.get("k_poa"))
P
}
for rep in 1 2; do
run "default"
MA_POA_WGS_PER_CU=1 run "wgs_per_cu=1"
MA_POA_MIN_FILLS=8 run "min_fills=8"
MA_POA_MIN_FILLS=32 run "min_fills=32"
MA_POA_PRIORITY=0 run "poa_priority=0"
MA_POA_ORDER=0 run "order=0"
done
