#!/bin/bash
set -u
O=gpurun_out/r6_poa_ab4
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_wide_components.py -x -q -m gpu -k "msa or wide or haplotypes" 2>&1 | tail -3
echo "== fill profile (lean rows incl. column-0 rows)" ; timeout 600 python3 tools/prof_phases.py 64 bench 2>&1 | grep "ma_debug_prof"
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
for m in 4 8 16 32 64; do run minfills${m}_1lane MA_POA_MIN_FILLS=$m MA_STREAMS=1; done
run sched0_4lanes MA_POA_SCHED=0
for m in 8 16 32; do run minfills${m}_4lanes MA_POA_MIN_FILLS=$m; done
run minfills16_4lanes_1wg MA_POA_MIN_FILLS=16 MA_POA_WGS_PER_CU=1
