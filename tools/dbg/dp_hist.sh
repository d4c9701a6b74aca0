# Developer tool: DP pairs by class and (libmicroasm_hist.so, -DMA_DP_HIST) by region width
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
export MA_BENCH_CACHE=/tmp/mbc
python3 bench.py --no-cpu --no-also --gen-only > /dev/null 2>&1
MA_LIB=$PWD/lancet2_amd/libmicroasm_hist.so MA_VOTE_DEBUG=1 MA_STREAMS=1 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-also 2>&1 | grep "DP pairs" | head -3 > gpurun_out/r5b/dp_hist.txt
