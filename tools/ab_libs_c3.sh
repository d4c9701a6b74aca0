#!/bin/bash
# developer tool: same-box A/B of library builds on the default (C3) workload.  usage: tools/ab_libs_c3.sh kernel lib1.so lib2.so ...
k=$1; shift
for lib in "$@"; do
  MA_LIB=$PWD/lancet2_amd/$lib MA_STREAMS=1 python3 bench.py --steps 3 --no-cpu --no-also 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'single-lane', d['value'], '$k', d['kernel_ms_per_step'].get('$k'))"
  MA_LIB=$PWD/lancet2_amd/$lib python3 bench.py --steps 4 --no-cpu --no-also 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'lanes', d['value'], d['ms_per_step'])"
done
