#!/bin/bash
# Developer tool (GPU box): k_poa with per-XCD hand-over against the host-counted rounds; parity of the POA tests first.
set -u
O=gpurun_out/r6_poa_ab2
mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "msa" 2>&1 | tail -4
MA_VERBOSE=1 timeout 300 python3 tools/poa_bench.py 8192 256 2>&1 | grep "k_poa\|sum:" | tail -4
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
          {x: k[x] for x in k if x.startswith(("k_poa", "k_msa"))}, "parity", d.get("parity_sample"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run sched0_4lanes MA_POA_SCHED=0
run sched1_4lanes MA_POA_SCHED=1
run sched1_4lanes_1wg MA_POA_SCHED=1 MA_POA_WGS_PER_CU=1
run sched0_1lane MA_POA_SCHED=0 MA_STREAMS=1
run sched1_1lane MA_POA_SCHED=1 MA_STREAMS=1
run sched1_1lane_1wg MA_POA_SCHED=1 MA_STREAMS=1 MA_POA_WGS_PER_CU=1
run sched1_1lane_devscope MA_POA_SCHED=1 MA_STREAMS=1 MA_POA_XCD=0
