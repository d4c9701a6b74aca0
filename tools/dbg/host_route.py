"""Developer tool: one MA_MEM_HOST ma_process_batch over pinned caller buffers, per-lane phase times (MA_VERBOSE)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from lancet2_amd import capi, synth
from lancet2_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
arrs, n0, nr0 = bench.make_windows("C3", 512, 10_000, 8, 8)
arrs, n, nr = synth.tile_batch(arrs, n0, nr0, n // 512)
params = capi.default_params(min_k=25, max_k=25)
keep = []
def pinned(a):
    t = torch.empty(max(a.nbytes, 16), dtype=torch.uint8, pin_memory=True); keep.append(t)
    v = t.numpy()[:a.nbytes].view(a.dtype); v[...] = a; return v
h_in = {k: pinned(np.ascontiguousarray(v)) for k, v in arrs.items()}
def pout(spec):
    return {k: pinned(np.zeros(int(sz), dtype=dt)) for k, (dt, sz) in spec.items()}
outs = (pout(capi.gate_out_spec(n)), pout(capi.asm_out_spec(params, n)), pout(capi.var_out_spec(params, n)), pout(capi.geno_out_spec(params, n, nr, debug=False)))
st = (capi.fill_struct(capi.GateOut, outs[0]), capi.fill_struct(capi.AsmOut, outs[1]), capi.fill_struct(capi.VarOut, outs[2]), capi.fill_struct(capi.GenoOut, outs[3]))
b = capi.make_batch_struct(h_in, n, nr)
eng = Engine(params, memspace=capi.MA_MEM_HOST)
eng.timing_control(0)
eng.process_device(b, *st)
eng.prefetch(b)
time.sleep(0.1)
os.environ["MA_VERBOSE"] = "1"
t = time.perf_counter(); eng.process_device(b, *st); dt = time.perf_counter() - t
os.environ.pop("MA_VERBOSE")
eng.prefetch(b)
t = time.perf_counter()
for _ in range(4):
    eng.prefetch(b)
    eng.process_device(b, *st)
dt3 = (time.perf_counter() - t) / 4
print("batch", n, "windows:", round(dt * 1e3, 1), "ms verbose;", round(dt3 * 1e3, 1), "ms ->", round(n / dt3), "windows/s")
eng.close()
