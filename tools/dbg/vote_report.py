"""Developer tool: print what tools/dbg/vote_iter.sh brought back."""
import json, ast
print(open("gpurun_out/r5b/gpu_tests_align.txt").read().strip().splitlines()[-1])
L = open("gpurun_out/r5b/vprof.txt").read().strip().splitlines()
kt = ast.literal_eval(L[0]); v = ast.literal_eval(L[1].split(" ", 1)[1])
trips = max(v["pairs"], 1)
print("prof build k_vote ms/4096:", kt.get("k_vote"), "| ticks per trip: cand %.0f  group vote %.0f  wave-wide %.0f  total %.0f  (trips %d, set-up %.0f per WG-thread0)" % (
    v["encode"] / trips, v["vote"] / trips, v["argmax"] / trips, v["wave_total"] / trips, trips, v["wg_setup(thread 0)"]))
for f in ("bench_1lane", "bench"):
    d = json.load(open(f"gpurun_out/r5b/{f}.json"))
    print(f, d["value"], d["ms_per_step"], "k_vote", d["kernel_ms_per_step"].get("k_vote"), "k_align_reg", d["kernel_ms_per_step"].get("k_align_reg"), "parity", d.get("parity_sample"))
gt = max(v["shortcut"], 1)
print("group vote: %d trips of eight listed reads; ticks per such trip: total %.0f = lookup %.0f  reduce+mismatch %.0f  settle %.0f  queue+list %.0f" % (
    gt, v["vote"] / gt, v["second"] / gt, v["clear"] / gt, v["mismatch"] / gt, v["emit"] / gt))
