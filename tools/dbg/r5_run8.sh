cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
export MA_BENCH_CACHE=/tmp/mbc
python3 bench.py --no-cpu --no-also --gen-only > /dev/null 2>&1
for e in 0 44 0 44; do
  if [ $e != 0 ]; then export MA_WS_GB=$e; else unset MA_WS_GB; fi
  python3 bench.py --steps 4 --no-cpu --no-also 2>gpurun_out/r5b/ws_err.txt | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ws_gb=$e', 'lanes', d.get('value'), d.get('ms_per_step'), d.get('error'))" >> gpurun_out/r5b/ab_ws.txt
done
