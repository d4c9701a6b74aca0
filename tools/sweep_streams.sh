#!/bin/bash
# developer tool: throughput of bench.py for 1..4 concurrent window ranges (same box, back to back)
for s in 1 2 3 4; do
  MA_STREAMS=$s python3 bench.py --steps 4 --no-cpu --no-also 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams', $s, d['value'], d['ms_per_step'])"
done
