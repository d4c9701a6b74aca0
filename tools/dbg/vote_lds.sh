cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
export MA_BENCH_CACHE=/tmp/mbc
python3 bench.py --no-cpu --no-also --gen-only > /dev/null 2>&1
MA_VOTE_DEBUG=1 MA_STREAMS=1 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-also 2>&1 | grep "k_vote:" | head -12 > gpurun_out/r5b/vote_lds.txt
