#!/bin/bash
# Developer tool (GPU box): round-6 first look -- the N > 1 rehearsal test, a short bench line with the gcups block, and the
# POA rounds of the bench workload (per-launch times; in-kernel fills for comparison).
set -u
O=gpurun_out/r6_diag1
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_multirank.py tests/test_gpu_aligner.py -x -q -m gpu > $O/tests.txt 2>&1
tail -5 $O/tests.txt
export MA_BENCH_CACHE=/tmp/ma_bench_cache
timeout 900 python3 bench.py --no-cpu --no-also --steps 4 2> $O/bench.err | tail -1 > $O/bench_short.json
MA_STREAMS=1 timeout 900 python3 bench.py --no-cpu --no-also --steps 4 2>> $O/bench.err | tail -1 > $O/bench_short_single_lane.json
python3 - <<'PY'
import json
for f in ("bench_short.json", "bench_short_single_lane.json"):
    try:
        d = json.load(open("gpurun_out/r6_diag1/" + f))
        print(f, d["value"], d["ms_per_step"], json.dumps(d.get("gcups"))[:1500])
        print({k: v for k, v in list(d["kernel_ms_per_step"].items())[:12]})
    except Exception as e:
        print(f, "failed", e)
PY
timeout 600 python3 tools/dbg/msa_rounds.py 8192 > $O/msa_rounds.txt 2>&1
tail -60 $O/msa_rounds.txt
MA_POA_BAND=1 timeout 600 python3 tools/dbg/msa_rounds.py 8192 > $O/msa_rounds_inkernel.txt 2>&1
tail -8 $O/msa_rounds_inkernel.txt
