# Developer tool: GPU idle gaps of the SINGLE-lane bench (what a lane's own chain of launches and host round trips costs)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
export MA_BENCH_CACHE=/tmp/mbc
python3 bench.py --no-cpu --no-also --gen-only > /dev/null 2>&1
rm -rf gpurun_out/r5b/ktr
MA_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r5b/ktr -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-also > gpurun_out/r5b/ktr.log 2>&1
python3 tools/dbg/trace_gaps.py gpurun_out/r5b/ktr 150 5 > gpurun_out/r5b/gaps1.txt 2>&1
rm -rf gpurun_out/r5b/ktr
