# Developer tool: the four-lane bench N times on cached windows (run-to-run spread of one library)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
export MA_BENCH_CACHE=/tmp/mbc
python3 bench.py --no-cpu --no-also --gen-only > /dev/null 2>&1
rm -f gpurun_out/r5b/bench_reps.txt
for rep in 1 2 3; do
  python3 bench.py --steps 4 --no-cpu --no-also 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(d['value'], d['ms_per_step'], 'k_align_reg', k.get('k_align_reg'), 'k_vote', k.get('k_vote'))" >> gpurun_out/r5b/bench_reps.txt
done
