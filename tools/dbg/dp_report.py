"""Developer tool: print what tools/dbg/dp_iter.sh brought back."""
import json
print(open("gpurun_out/r5b/gpu_tests_align.txt").read().strip().splitlines()[-1])
print(open("gpurun_out/r5b/dp_hist.txt").read().strip())
for f in ("bench_1lane", "bench"):
    d = json.load(open(f"gpurun_out/r5b/{f}.json")); k = d["kernel_ms_per_step"]
    print(f, d["value"], d["ms_per_step"], "k_align_reg", k.get("k_align_reg"), "k_align_tb", k.get("k_align_tb"), "k_vote", k.get("k_vote"), "parity", d.get("parity_sample"))
