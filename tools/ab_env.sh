#!/bin/bash
# developer tool: same-box A/B of an environment switch.  usage: tools/ab_env.sh kernel VAR val1 val2 ...
k=$1; var=$2; shift; shift
for v in "$@"; do
  export $var=$v
  MA_STREAMS=1 python3 bench.py --steps 3 --no-cpu 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$var=$v', 'single-lane', d['value'], {k_: v_ for k_, v_ in d['kernel_ms_per_step'].items() if k_.startswith('$k')})"
  python3 bench.py --steps 4 --no-cpu 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$var=$v', 'lanes', d['value'], d['ms_per_step'])"
done
