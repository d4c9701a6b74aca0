#!/bin/bash
# Developer tool: the measurement set behind profiles/r2_final_* (run on the GPU box through gpurun, from the repo root).
# PMC passes first: bench.py reads the dominant kernel's counters from profiles/r2_pmc_per_kernel.json.
set -u
R=$PWD
O=$R/gpurun_out/final
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --no-cpu --no-also --gen-workers 1"
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- $B > $O/pmc_$c.log 2>&1
done
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES \
  --kernel-trace --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
cd $R
python3 tools/pmc_per_kernel.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_INSTS_VALU profiles/r2_pmc_per_kernel.json > $O/pmc_per_kernel.txt 2>&1
cp profiles/r2_pmc_per_kernel.json $O/
grep -h "^{\"metric\"" $O/pmc_SQ_INSTS_VALU.log | tail -1 > $O/r2_pmc_bench_under_rocprof.json
python3 tools/dbg/pmc_generic.py $O/pmc_sq > $O/r2_pmc_sq_per_kernel.txt 2>&1
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -- python3 $R/bench.py --no-cpu --no-also --gen-workers 1 > $O/ktrace.log 2>&1
grep -h "^{\"metric\"" $O/ktrace.log | tail -1 > $O/r2_final_bench_under_rocprof.json
cp $(find $O/ktrace -name "*kernel_stats.csv" | head -1) $O/r2_final_kernel_stats.csv
cd $R
MA_STREAMS=1 timeout 300 python3 bench.py --no-cpu --no-also 2>&1 | tail -1 > $O/r2_final_bench_single_lane.json
MA_STREAMS=2 timeout 300 python3 bench.py --no-cpu --no-also 2>&1 | tail -1 > $O/bench_2_lanes.json
timeout 900 python3 bench.py 2>&1 | tail -1 > $O/r2_final_bench.json
# the big per-dispatch CSVs stay on the box
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_INSTS_VALU $O/pmc_sq $O/ktrace
for f in r2_final_bench.json r2_final_bench_single_lane.json bench_2_lanes.json r2_final_bench_under_rocprof.json; do
  python3 -c "import json,sys; d=json.load(open('$O/$f')); print('$f', d['value'], d['ms_per_step'])"
done
