cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r5c/gpu_tests.txt
timeout 1500 python bench.py > gpurun_out/r5c/bench.json 2> gpurun_out/r5c/bench.err
echo rc=$? >> gpurun_out/r5c/bench.err
