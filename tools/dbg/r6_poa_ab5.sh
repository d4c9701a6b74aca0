#!/bin/bash
set -u
O=gpurun_out/r6_poa_ab5
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_wide_components.py -x -q -m gpu -k "msa or wide or haplotypes" 2>&1 | tail -3
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
for f in 0 2 4 6 8 12; do run fill${f}_1lane MA_POA_FILL_WGS=$f MA_STREAMS=1; done
for f in 0 4 6 8; do run fill${f}_4lanes MA_POA_FILL_WGS=$f; done
run fill6_4lanes_1wg MA_POA_FILL_WGS=6 MA_POA_WGS_PER_CU=1
run fill6_4lanes_q16 MA_POA_FILL_WGS=6 GPU_MAX_HW_QUEUES=16
