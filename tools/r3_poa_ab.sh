#!/bin/bash
# Developer tool: POA tier A/B on the GPU box (POA stage alone, 8192 windows).
set -u
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "msa" 2>&1 | tail -3
for t in 1 2 4; do
  echo "== tier0=$t"
  MA_POA_TIER0=$t MA_VERBOSE=1 timeout 300 python3 tools/poa_bench.py 8192 256 2>&1 | grep -v "^\[microasm\] msa:" | tail -3
done
for t in 1 2 4; do
echo "== tier0=$t 2048 windows"
MA_POA_TIER0=$t timeout 300 python3 tools/poa_bench.py 2048 256 2>&1 | tail -2
done
echo "== not lean, tier0=4 8192"
MA_POA_LEAN=0 MA_POA_TIER0=4 timeout 300 python3 tools/poa_bench.py 8192 256 2>&1 | tail -2
