#!/bin/bash
set -u
O=gpurun_out/r6_poa_prof2
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "msa" 2>&1 | tail -2
for n in 64 8192; do
  echo "== $n windows (single lane)"
  timeout 600 python3 tools/prof_phases.py $n bench 2>&1 | grep "ma_debug_prof\|k_poa" | cut -c1-900
done
