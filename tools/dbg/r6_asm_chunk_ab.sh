#!/bin/bash
# Developer tool: one ASSEMBLY chunk per lane against two, at half scale (8192 windows per step: a lane's 2048 windows fit one
# chunk of the default budget; MA_WS_GB=11 cuts them in two -- the POA stage still takes its 2048 in one)
export MA_BENCH_CACHE=/tmp/ma_bench_cache
O=gpurun_out/r6_asmchunk
mkdir -p $O
python3 bench.py --no-cpu --no-also --gen-only --windows 8192 --distinct 8192 > $O/gen.log 2>&1
for rep in 1 2; do
for gb in 0 11; do
  if [ $gb = 0 ]; then unset MA_WS_GB; else export MA_WS_GB=$gb; fi
  timeout 300 python3 bench.py --no-cpu --no-also --gen-workers 1 --windows 8192 --distinct 8192 --steps 16 2> $O/err_$gb.txt | tail -1 > $O/b_$gb.json
  python3 - <<P
import json
try:
    d=json.load(open("$O/b_$gb.json"))
    k=d["kernel_ms_per_step"]
    print("ws_gb $gb rep $rep", d["value"], d["ms_per_step"], {x:k.get(x) for x in ("k_classify","k_insert","k_support","k_graph","k_clean_chains","k_clean_tail","k_poa")})
except Exception as e:
    print("ws_gb $gb failed", e); print(open("$O/err_$gb.txt").read()[-600:])
P
done
done
