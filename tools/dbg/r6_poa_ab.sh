#!/bin/bash
# Developer tool (GPU box): the persistent POA kernel (k_poa) against the host-counted rounds, four lanes and one, on one box.
set -u
O=gpurun_out/r6_poa_ab
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
          {x: k[x] for x in k if x.startswith(("k_poa", "k_msa"))}, "gcups poa", (d.get("gcups", {}).get("poa_band") or {}).get("GCUPS"))
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
run sched1_4lanes_b MA_POA_SCHED=1
run sched0_4lanes_b MA_POA_SCHED=0
