#!/bin/bash
# Developer tool: GPU idle gaps of the SINGLE-lane bench (where is the step that is not kernels?) and of the two-lane one
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_gaps1
mkdir -p $O
export MA_BENCH_CACHE=/tmp/mbc
python3 bench.py --no-cpu --no-also --gen-only > /dev/null 2>&1
for lanes in 1 2; do
  rm -rf $O/ktr
  MA_STREAMS=$lanes timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/ktr -- python3 bench.py --steps 4 --warmup 2 --no-cpu --no-also > $O/ktr_$lanes.log 2>&1
  python3 tools/dbg/trace_gaps.py $O/ktr 160 0.02 > $O/gaps_$lanes.txt 2>&1
  rm -rf $O/ktr
  echo "== lanes $lanes"; head -40 $O/gaps_$lanes.txt
done
