cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r5b/gpu_tests.txt
if grep -q failed gpurun_out/r5b/gpu_tests.txt; then exit 0; fi
timeout 600 python tools/dbg/cascade_passes.py 2048 2>/dev/null | grep -v "^\[" > gpurun_out/r5b/cascade.txt
export MA_BENCH_CACHE=/tmp/mbc
timeout 900 python bench.py --no-cpu > gpurun_out/r5b/bench.json 2> gpurun_out/r5b/bench.err
