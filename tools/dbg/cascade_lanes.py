"""Developer tool: the default k ladder on the device-resident route, four lanes: windows/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
import bench
from lancet2_amd import capi, synth
from lancet2_amd.engine import Engine
n = 8192
distinct = int(os.environ.get("DISTINCT", "2048"))
arrs, n0, nr0 = bench.make_windows("C3", distinct, 10_000, 8, 8)
arrs, n, nr = synth.tile_batch(arrs, n0, nr0, n // distinct)
params = capi.default_params(min_k=13, max_k=127, k_step=6)
dev = torch.device("cuda:0")
d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in arrs.items()}
def dout(spec):
    return {k: torch.zeros(int(sz) * np.dtype(dt).itemsize, dtype=torch.uint8, device=dev) for k, (dt, sz) in spec.items()}
outs = (dout(capi.gate_out_spec(n)), dout(capi.asm_out_spec(params, n)), dout(capi.var_out_spec(params, n)), dout(capi.geno_out_spec(params, n, nr, debug=False)))
st = (capi.fill_struct(capi.GateOut, outs[0]), capi.fill_struct(capi.AsmOut, outs[1]), capi.fill_struct(capi.VarOut, outs[2]), capi.fill_struct(capi.GenoOut, outs[3]))
b = capi.make_batch_struct(d_in, n, nr)
eng = Engine(params, memspace=capi.MA_MEM_DEVICE)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.timing_control(0)
for _ in range(2):
    eng.process_device(b, *st)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3):
    eng.process_device(b, *st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 3
print("k ladder 13..127, four lanes: %.1f ms/step -> %.0f submitted windows/s" % (dt * 1e3, n / dt), flush=True)
eng.close()
