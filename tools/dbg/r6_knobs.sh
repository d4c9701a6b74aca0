#!/bin/bash
# Developer tool (GPU box): the knobs around the persistent POA kernel under concurrent lanes (headline workload, 6 steps each)
set -u
O=gpurun_out/r6_knobs
mkdir -p $O
export MA_BENCH_CACHE=/tmp/ma_bench_cache
python3 bench.py --no-cpu --no-also --gen-only > $O/gen.log 2>&1
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout 600 python3 bench.py --no-cpu --no-also --steps 6 2>> $O/err.txt | tail -1 > $O/$label.json
  python3 - "$O/$label.json" "$label" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = d["kernel_ms_per_step"]
    print("%-28s %9.1f w/s %7.2f ms/step  poa:" % (sys.argv[2], d["value"], d["ms_per_step"]),
          {x: k[x] for x in k if x.startswith(("k_poa", "k_msa"))})
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run base X=1
run prio0 MA_POA_PRIORITY=0
run wgs1 MA_POA_WGS_PER_CU=1
run wgs1_prio0 MA_POA_WGS_PER_CU=1 MA_POA_PRIORITY=0
run lanes3 MA_STREAMS=3
run lanes5 MA_STREAMS=5
run lanes6 MA_STREAMS=6
run base_b X=1
