set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r5a/gpu_tests.txt
timeout 900 python tools/dbg/c4_flags.py 512 > gpurun_out/r5a/c4_flags.txt 2>&1
timeout 1500 python bench.py > gpurun_out/r5a/bench.json 2> gpurun_out/r5a/bench.err
echo rc=$? >> gpurun_out/r5a/bench.err
