#!/bin/bash
# Developer tool: the number of lanes once a lane's windows go through every stage in one chunk
export MA_BENCH_CACHE=/tmp/ma_bench_cache
O=gpurun_out/r6_lanes2
mkdir -p $O
python3 bench.py --no-cpu --no-also --gen-only > $O/gen.log 2>&1
for rep in 1 2; do
  for lanes in 2 3 4 5 6; do
    MA_STREAMS=$lanes timeout 300 python3 bench.py --no-cpu --no-also --gen-workers 1 2>$O/err.txt | tail -1 > $O/b_${lanes}_$rep.json
    python3 - <<P
import json
try:
    d=json.load(open("$O/b_${lanes}_$rep.json"))
    print("lanes $lanes rep $rep", d["value"], d["ms_per_step"])
except Exception as e:
    print("lanes $lanes failed", open("$O/err.txt").read()[-300:])
P
  done
done
