#!/bin/bash
# Developer tool: the host route (MA_MEM_HOST) of bench.py's `also` legs, new against legacy
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "host_route" 2>&1 | tail -2
for v in "X=1" "MA_HOST_LEGACY=1" "MA_STREAMS=2" "MA_STREAMS=3"; do
  echo "== $v"
  env $v python3 bench.py --no-cpu --steps 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); a=d['also']
print('device', d['value'], d['config']['submitted_windows_per_s'], 'host1', a['host_path'].get('value'), 'host2', a.get('host_path_2_feeders',{}).get('value'), a['host_path'].get('error'))"
done
