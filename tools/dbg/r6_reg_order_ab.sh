#!/bin/bash
# Developer tool: k_align_reg2p's workgroups heaviest first (the library) against list order (libmicroasm_order0.so: -DMA_REG_ORDER=0)
export MA_BENCH_CACHE=/tmp/ma_bench_cache
O=gpurun_out/r6_regorder
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_aligner.py tests/test_gpu_parity.py -x -q -m gpu -k "align or geno or process or pairs" 2>&1 | tail -2
python3 bench.py --no-cpu --no-also --gen-only > $O/gen.log 2>&1
run() {
  timeout 300 python3 bench.py --no-cpu --no-also --gen-workers 1 2>$O/err.txt | tail -1 > $O/b.json
  python3 - <<P
import json
d=json.load(open("$O/b.json"))
k=d["kernel_ms_per_step"]
print("$1", d["value"], d["ms_per_step"], "k_align_reg", k.get("k_align_reg"), d["parity_sample"])
P
}
for rep in 1 2 3; do
run "heaviest first"
MA_LIB=$PWD/lancet2_amd/libmicroasm_order0.so run "list order"
MA_STREAMS=1 run "heaviest first, 1 lane"
MA_STREAMS=1 MA_LIB=$PWD/lancet2_amd/libmicroasm_order0.so run "list order, 1 lane"
done
