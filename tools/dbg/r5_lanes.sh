cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
for L in 2 3 4 6 8; do
  for Q in 8 16; do
    GPU_MAX_HW_QUEUES=$Q MA_STREAMS=$L timeout 300 python bench.py --no-cpu --no-also --steps 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('lanes $L queues $Q', d['value'], d['ms_per_step'])" >> gpurun_out/r5f/lanes.txt
  done
done
MA_STREAMS=4 timeout 300 python bench.py --no-cpu --no-also --steps 4 --windows 16384 --distinct 16384 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('lanes 4 16384 windows', d['value'], d['ms_per_step'])" >> gpurun_out/r5f/lanes.txt
