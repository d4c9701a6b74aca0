#!/bin/bash
# Developer tool: per-lane busy / idle time of the four-lane bench (kernel trace -> tools/dbg/trace_lanes.py, trace_gaps.py)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_gaps
mkdir -p $O
export MA_BENCH_CACHE=/tmp/mbc
python3 bench.py --no-cpu --no-also --gen-only > /dev/null 2>&1
rm -rf $O/ktr
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/ktr -- python3 bench.py --steps 4 --warmup 2 --no-cpu --no-also > $O/ktr.log 2>&1
python3 tools/dbg/trace_lanes.py $O/ktr 150 > $O/lanes.txt 2>&1
python3 tools/dbg/trace_gaps.py $O/ktr 150 > $O/gaps.txt 2>&1
rm -rf $O/ktr
head -70 $O/lanes.txt
