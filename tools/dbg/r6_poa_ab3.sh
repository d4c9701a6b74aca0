#!/bin/bash
set -u
O=gpurun_out/r6_poa_ab3
mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "msa_persistent or msa_parity_band" 2>&1 | tail -3
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
run sched0_1lane MA_POA_SCHED=0 MA_STREAMS=1
run minfills1_1lane MA_POA_MIN_FILLS=1 MA_STREAMS=1
run minfills4_1lane MA_POA_MIN_FILLS=4 MA_STREAMS=1
run minfills16_1lane MA_POA_MIN_FILLS=16 MA_STREAMS=1
run minfills100000_1lane MA_POA_MIN_FILLS=100000 MA_STREAMS=1
run sched0_4lanes MA_POA_SCHED=0
run minfills4_4lanes MA_POA_MIN_FILLS=4
run minfills100000_4lanes MA_POA_MIN_FILLS=100000
