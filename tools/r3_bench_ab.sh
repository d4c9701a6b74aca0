#!/bin/bash
# Developer tool: bench A/B (single lane + default lanes) with the env given on the command line
set -u
for v in "$@"; do
  echo "== $v"
  env $v MA_STREAMS=1 timeout 300 python3 bench.py --no-cpu --no-also --steps 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('single-lane', d['value'], d['ms_per_step'], {x:k[x] for x in k if 'msa' in x})"
  env $v timeout 300 python3 bench.py --no-cpu --no-also --steps 5 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('lanes', d['value'], d['ms_per_step'], {x:k[x] for x in k if 'msa' in x})"
done
