cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r5b/gpu_tests.txt
if grep -q failed gpurun_out/r5b/gpu_tests.txt; then exit 0; fi
timeout 600 python bench.py --no-cpu --no-also > gpurun_out/r5b/bench.json 2> gpurun_out/r5b/bench.err
MA_STREAMS=1 timeout 600 python bench.py --no-cpu --no-also > gpurun_out/r5b/bench_1lane.json 2>> gpurun_out/r5b/bench.err
