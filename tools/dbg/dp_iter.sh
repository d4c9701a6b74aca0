# Developer tool: one iteration on the DP-region work -- aligner/genotype parity tests, region-width histogram, bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests -m gpu -x -q -k "align or geno or vote or parity" 2>&1 | tail -5 > gpurun_out/r5b/gpu_tests_align.txt
export MA_BENCH_CACHE=/tmp/mbc
python3 bench.py --no-cpu --no-also --gen-only > /dev/null 2>&1
MA_LIB=$PWD/lancet2_amd/libmicroasm_hist.so MA_VOTE_DEBUG=1 MA_STREAMS=1 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-also 2>&1 | grep "DP pairs" | head -1 > gpurun_out/r5b/dp_hist.txt
MA_STREAMS=1 python3 bench.py --steps 3 --no-cpu --no-also 2>/dev/null | tail -1 > gpurun_out/r5b/bench_1lane.json
python3 bench.py --steps 4 --no-cpu --no-also 2>/dev/null | tail -1 > gpurun_out/r5b/bench.json
