#!/bin/bash
# Developer tool: one arena for the assembly and POA stages (a lane's 4096 windows in ONE chunk each): chunks, then the bench
export MA_BENCH_CACHE=/tmp/ma_bench_cache
O=gpurun_out/r6_onechunk
mkdir -p $O
python3 bench.py --no-cpu --no-also --gen-only > $O/gen.log 2>&1
MA_VERBOSE=1 timeout 300 python3 bench.py --no-cpu --no-also --gen-workers 1 --steps 1 --warmup 1 2>&1 | grep "msa:\|assemble:" | cut -c1-150 | sort | uniq -c | sort -rn | head -6
for rep in 1 2 3; do
  for lanes in 4 1; do
    MA_STREAMS=$lanes timeout 300 python3 bench.py --no-cpu --no-also --gen-workers 1 2>$O/err.txt | tail -1 > $O/b_${lanes}_$rep.json
    python3 - <<P
import json
try:
    d=json.load(open("$O/b_${lanes}_$rep.json"))
    k=d["kernel_ms_per_step"]
    print("lanes $lanes rep $rep", d["value"], d["ms_per_step"], {x:k.get(x) for x in ("k_classify","k_insert","k_support","k_graph","k_clean_chains","k_clean_tail","k_poa")})
except Exception as e:
    print("failed", e); print(open("$O/err.txt").read()[-800:])
P
  done
done
rocm-smi --showmeminfo vram 2>/dev/null | tail -3
