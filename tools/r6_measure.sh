#!/bin/bash
# Developer tool: the measurement set behind profiles/r6_* (run on the GPU box through gpurun, from the repo root).
# usage: tools/r6_measure.sh <tag>   -> gpurun_out/<tag>/...   (PMC passes first: bench.py reads profiles/r6_pmc_per_kernel.json)
set -u
R=$PWD
T=${1:-r6_final}
O=$R/gpurun_out/$T
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# the windows are synthesised ONCE (16 processes) and kept: every profiler pass below loads them
export MA_BENCH_CACHE=/tmp/ma_bench_cache
python3 $R/bench.py --no-cpu --no-also --gen-only > $O/gen.log 2>&1
B="python3 $R/bench.py --steps 2 --no-cpu --no-also --gen-workers 1"
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- $B > $O/pmc_$c.log 2>&1
done
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES \
  --kernel-trace --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
cd $R
python3 tools/pmc_per_kernel.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_INSTS_VALU $O/${T}_pmc_per_kernel.json $O/pmc_SQ_INSTS_VALU.log > $O/${T}_pmc_per_kernel.txt 2>&1
cp $O/${T}_pmc_per_kernel.json profiles/r6_pmc_per_kernel.json
grep -h "^{\"metric\"" $O/pmc_SQ_INSTS_VALU.log | tail -1 > $O/${T}_pmc_bench_under_rocprof.json
python3 tools/dbg/pmc_generic.py $O/pmc_sq - $O/${T}_pmc_sq_per_kernel.json > $O/${T}_pmc_sq_per_kernel.txt 2>&1
cp $O/${T}_pmc_sq_per_kernel.json profiles/r6_pmc_sq_per_kernel.json
# achieved occupancy and L2 hit rate, each in a pass of its own (a pass that asks for too much aborts the profiler)
cd /tmp
timeout 600 rocprofv3 --pmc SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/pmc_occ -- $B > $O/pmc_occ.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_l2 -- $B > $O/pmc_l2.log 2>&1
cd $R
python3 tools/dbg/pmc_generic.py $O/pmc_occ > $O/${T}_pmc_occupancy_per_kernel.txt 2>&1
python3 tools/dbg/pmc_generic.py $O/pmc_l2 > $O/${T}_pmc_l2_per_kernel.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -- python3 $R/bench.py --no-cpu --no-also --gen-workers 1 > $O/ktrace.log 2>&1
grep -h "^{\"metric\"" $O/ktrace.log | tail -1 > $O/${T}_bench_under_rocprof.json
cp $(find $O/ktrace -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats.csv
cd $R
MA_STREAMS=1 timeout 300 python3 bench.py --no-cpu --no-also 2>/dev/null | tail -1 > $O/${T}_bench_single_lane.json
if [ "${2:-}" = "full" ]; then timeout 1200 python3 bench.py 2>/dev/null | tail -1 > $O/${T}_bench.json; fi
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_INSTS_VALU $O/pmc_sq $O/pmc_occ $O/pmc_l2 $O/ktrace
cat $O/${T}_pmc_per_kernel.txt | sort -k5 -n -r | head -30
