cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
timeout 1500 python bench.py > gpurun_out/r5c/bench.json 2> gpurun_out/r5c/bench.err
echo rc=$? >> gpurun_out/r5c/bench.err
