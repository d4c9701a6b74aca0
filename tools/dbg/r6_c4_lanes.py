"""Developer tool (GPU box): the deep-panel leg (C4, 2048 windows per step, 512 distinct) under 2 / 4 / 6 / 8 concurrent lanes.
usage: python3 tools/dbg/r6_c4_lanes.py [distinct=256]"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from lancet2_amd import capi, synth  # noqa: E402
from lancet2_amd.engine import Engine  # noqa: E402

distinct = int(sys.argv[1]) if len(sys.argv) > 1 else 256
arrs, nw, nr = bench.make_windows("C4", distinct, 10_000, 0, 16)
arrs, nw, nr = synth.tile_batch(arrs, nw, nr, 2048 // distinct)
dev = torch.device("cuda", 0)
d = {k: torch.from_numpy(v.view(np.uint8) if v.dtype != np.uint8 else v).to(dev) for k, v in arrs.items()}
b = capi.make_batch_struct(d, nw, nr)
p = capi.default_params(min_k=25, max_k=25)


def alloc(spec):
    return {k: torch.zeros(int(sz) * np.dtype(dt).itemsize, dtype=torch.uint8, device=dev) for k, (dt, sz) in spec.items()}


g, a, v, q = alloc(capi.gate_out_spec(nw)), alloc(capi.asm_out_spec(p, nw)), alloc(capi.var_out_spec(p, nw)), alloc(capi.geno_out_spec(p, nw, nr, debug=False))
st = (capi.fill_struct(capi.GateOut, g), capi.fill_struct(capi.AsmOut, a), capi.fill_struct(capi.VarOut, v), capi.fill_struct(capi.GenoOut, q))
for lanes in (2, 4, 6, 8, 2):
    eng = Engine(p, device=0, memspace=capi.MA_MEM_DEVICE)
    eng.set_streams(lanes)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    eng.timing_control(0)
    eng.process_device(b, *st)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(2):
        eng.process_device(b, *st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 2
    print("lanes %d: %.1f ms per step of %d windows = %.0f submitted windows/s" % (lanes, dt * 1e3, nw, nw / dt), flush=True)
    eng.close()
