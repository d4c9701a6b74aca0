# Developer tool: aligner / genotype parity tests + the single-lane and four-lane bench on cached windows
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests -m gpu -x -q -k "align or geno or vote or parity" 2>&1 | tail -5 > gpurun_out/r5b/gpu_tests_align.txt
export MA_BENCH_CACHE=/tmp/mbc
python3 bench.py --no-cpu --no-also --gen-only > /dev/null 2>&1
MA_STREAMS=1 python3 bench.py --steps 4 --no-cpu --no-also 2>/dev/null | tail -1 > gpurun_out/r5b/bench_1lane.json
python3 bench.py --no-cpu --no-also 2>/dev/null | tail -1 > gpurun_out/r5b/bench.json
