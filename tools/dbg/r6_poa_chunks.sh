#!/bin/bash
# Developer tool: what chunks the POA stage runs in (MA_VERBOSE) with 4 lanes and with 1
export MA_BENCH_CACHE=/tmp/ma_bench_cache
mkdir -p gpurun_out/r6_chunks
python3 bench.py --no-cpu --no-also --gen-only > gpurun_out/r6_chunks/gen.log 2>&1
for lanes in 4 1; do
  MA_VERBOSE=1 MA_STREAMS=$lanes timeout 300 python3 bench.py --no-cpu --no-also --gen-workers 1 --steps 1 --warmup 1 2> gpurun_out/r6_chunks/v_$lanes.txt | tail -1 | cut -c1-200
  grep "msa:\|k_poa:" gpurun_out/r6_chunks/v_$lanes.txt | sort | uniq -c | sort -rn | head -8
done
