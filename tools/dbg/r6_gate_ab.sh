#!/bin/bash
# Developer tool: the repeat gate with two diagonals per lane in packed halves (the library) against one (libmicroasm_gate0.so: -DMA_GATE_PACKED=0)
export MA_BENCH_CACHE=/tmp/ma_bench_cache
O=gpurun_out/r6_gate
mkdir -p $O
python3 bench.py --no-cpu --no-also --gen-only > $O/gen.log 2>&1
run() {
  timeout 300 python3 bench.py --no-cpu --no-also --gen-workers 1 2>$O/err.txt | tail -1 > $O/b.json
  python3 - <<P
import json
d=json.load(open("$O/b.json"))
k=d["kernel_ms_per_step"]
print("$1", d["value"], d["ms_per_step"], "gate", k.get("gate_kernel"), d["parity_sample"])
P
}
for rep in 1 2 3; do
run "packed"
MA_LIB=$PWD/lancet2_amd/libmicroasm_gate0.so run "one diagonal per lane"
MA_STREAMS=1 run "packed, 1 lane"
MA_STREAMS=1 MA_LIB=$PWD/lancet2_amd/libmicroasm_gate0.so run "one per lane, 1 lane"
done
