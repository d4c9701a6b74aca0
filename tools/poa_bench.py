"""Developer tool: the POA stage alone (ma_msa_batch on engine-assembled C3 windows), every kernel launch timed.
usage: python tools/poa_bench.py [windows=8192] [distinct=256]   (env: MA_LIB, MA_POA_TIER0, MA_POA_MIN_PENDING, ...)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lancet2_amd import capi, synth  # noqa: E402
from lancet2_amd import engine as E  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
distinct = int(sys.argv[2]) if len(sys.argv) > 2 else 256
arrs, nw, nr = synth.make_config_batch("C3", distinct, first_index=10_000)
arrs, nw, nr = synth.tile_batch(arrs, nw, nr, max(1, n // distinct))
eng = E.Engine(capi.default_params(min_k=25, max_k=25))
asm = eng.assemble(arrs, nw, nr)
eng.msa(arrs, nw, nr, asm)
eng.timing_control(1)
t0 = time.perf_counter()
eng.msa(arrs, nw, nr, asm)
wall = time.perf_counter() - t0
kt = eng.kernel_times()
agg = {}
for k, v in kt:
    agg[k] = agg.get(k, 0.0) + v
print("launches:", " ".join(f"{k.replace('k_msa', 'm')}={v:.2f}" for k, v in kt))
print("sum:", {k: round(v, 2) for k, v in agg.items()}, "total", round(sum(agg.values()), 2), "ms; wall incl. copies", round(wall * 1e3, 1))
eng.close()
