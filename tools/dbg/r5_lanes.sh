# Developer tool: the bench at 16384 windows per step with 3 ... 8 lanes (MA_STREAMS)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
export MA_BENCH_CACHE=/tmp/mbc
python3 bench.py --no-cpu --no-also --gen-only > /dev/null 2>&1
rm -f gpurun_out/r5f/lanes2.txt
for L in 3 4 5 6 8 4; do
  MA_STREAMS=$L timeout 300 python bench.py --no-cpu --no-also --steps 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('16384 windows, lanes $L', d['value'], d['ms_per_step'])" >> gpurun_out/r5f/lanes2.txt
done
