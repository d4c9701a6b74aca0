cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests -m gpu -x -q -k "deep or fused_graph or capacity or table" 2>&1 | tail -4 > gpurun_out/r5b/t.txt
export MA_BENCH_CACHE=/tmp/mbc
timeout 900 python bench.py --config C4 --windows 2048 --distinct 512 --steps 2 --warmup 1 --no-cpu --no-also --str-every 0 --hard-every 0 2>/dev/null | tail -1 > gpurun_out/r5b/c4.json
