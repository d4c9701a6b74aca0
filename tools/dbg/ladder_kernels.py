"""Developer tool: kernel times of the default ladder (k = 13 ... 127) on the bench workload, one lane and the default lanes.
usage: python tools/dbg/ladder_kernels.py [windows]   (MA_BENCH_CACHE as for bench.py)"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from lancet2_amd import capi  # noqa: E402
from lancet2_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
arrs, nw, nr = bench.make_windows("C3", n, 10_000, 8, 16)
dev = torch.device("cuda", 0)
d = {k: torch.from_numpy(v.view(np.uint8) if v.dtype != np.uint8 else v).to(dev) for k, v in arrs.items()}
b = capi.make_batch_struct(d, nw, nr)
p = capi.default_params()


def alloc(spec):
    return {k: torch.zeros(int(sz) * np.dtype(dt).itemsize, dtype=torch.uint8, device=dev) for k, (dt, sz) in spec.items()}


g, a, v, q = alloc(capi.gate_out_spec(nw)), alloc(capi.asm_out_spec(p, nw)), alloc(capi.var_out_spec(p, nw)), alloc(capi.geno_out_spec(p, nw, nr, debug=False))
st = (capi.fill_struct(capi.GateOut, g), capi.fill_struct(capi.AsmOut, a), capi.fill_struct(capi.VarOut, v), capi.fill_struct(capi.GenoOut, q))
for lanes in (1, 4):
    eng = Engine(p, device=0, memspace=capi.MA_MEM_DEVICE)
    eng.set_streams(lanes)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    eng.timing_control(0)
    eng.process_device(b, *st)
    torch.cuda.synchronize()
    eng.timing_control(2)
    t = time.perf_counter()
    for _ in range(2):
        eng.process_device(b, *st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 2
    kt = {}
    for name, ms in eng.kernel_times():
        kt[name] = kt.get(name, 0.0) + ms / 2
    print("lanes %d: %.1f ms per step; kernels summed %.1f ms:" % (lanes, dt * 1e3, sum(kt.values())),
          ", ".join("%s %.1f" % (k, x) for k, x in sorted(kt.items(), key=lambda y: -y[1])[:18]))
    eng.close()
