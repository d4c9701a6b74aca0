cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
export MA_BENCH_CACHE=/tmp/mbc
python3 bench.py --no-cpu --no-also --gen-only > /dev/null 2>&1
for L in 4 5 6 8; do
  MA_STREAMS=$L timeout 300 python bench.py --no-cpu --no-also --steps 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('16384 windows, lanes $L', d['value'], d['ms_per_step'])" >> gpurun_out/r5f/lanes2.txt
done
MA_STREAMS=8 timeout 300 python bench.py --no-cpu --no-also --steps 3 --windows 32768 --distinct 16384 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('32768 windows (16384 distinct x2), lanes 8', d['value'], d['ms_per_step'])" >> gpurun_out/r5f/lanes2.txt
MA_STREAMS=4 timeout 300 python bench.py --no-cpu --no-also --steps 3 --windows 32768 --distinct 16384 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('32768 windows (16384 distinct x2), lanes 4', d['value'], d['ms_per_step'])" >> gpurun_out/r5f/lanes2.txt
