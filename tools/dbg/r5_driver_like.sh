cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5h
unset MA_BENCH_CACHE
S=$(date +%s)
python bench.py > gpurun_out/r5h/bench.json 2> gpurun_out/r5h/bench.err
echo rc=$? elapsed=$(( $(date +%s) - S )) >> gpurun_out/r5h/bench.err
