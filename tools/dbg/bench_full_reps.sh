# Developer tool: the driver's command (python bench.py, all legs) twice, then the no-cpu/no-also variant -- is the full run slower?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
export MA_BENCH_CACHE=/tmp/mbc
rm -f gpurun_out/r5b/bench_full_reps.txt
for rep in 1 2; do
  python3 bench.py --no-also 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('with cpu legs', d['value'], d['ms_per_step'], d['steps'], d['warmup'], d['parity_sample']['mismatches'])" >> gpurun_out/r5b/bench_full_reps.txt
  python3 bench.py --no-cpu --no-also 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('no cpu legs  ', d['value'], d['ms_per_step'], d['steps'], d['warmup'])" >> gpurun_out/r5b/bench_full_reps.txt
done
