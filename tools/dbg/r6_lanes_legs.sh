#!/bin/bash
# Developer tool: every leg of the bench with two lanes against the automatic four
export MA_BENCH_CACHE=/tmp/ma_bench_cache
O=gpurun_out/r6_lanes_legs
mkdir -p $O
for lanes in 2 0; do
  if [ $lanes = 0 ]; then unset MA_STREAMS; else export MA_STREAMS=$lanes; fi
  timeout 900 python3 bench.py --no-cpu 2>$O/err_$lanes.txt | tail -1 > $O/b_$lanes.json
  python3 - <<P
import json
d=json.load(open("$O/b_$lanes.json"))
a=d["also"]
def val(v):
    if isinstance(v,dict):
        for k in ("assembled_windows_per_s","windows_per_s","value","submitted_windows_per_s"):
            if k in v: return v[k]
        return {k:x for k,x in v.items() if isinstance(x,(int,float))}
    return v
print("lanes $lanes headline", d["value"], d["ms_per_step"])
for k,v in a.items(): print("   ", k, str(val(v))[:200])
P
done
