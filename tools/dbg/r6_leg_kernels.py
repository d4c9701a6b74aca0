"""Developer tool (GPU box): kernel times of one of bench.py's secondary legs, one lane and the default lanes.
usage: python3 tools/dbg/r6_leg_kernels.py reads250|c5|c2 [windows]"""
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

leg = sys.argv[1] if len(sys.argv) > 1 else "reads250"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
p = capi.default_params(min_k=25, max_k=25)
if leg == "reads250":
    arrs, nw, nr = bench.make_windows("C3", 512, 10_000, 8, 16, over=dict(read_len=250))
elif leg == "c5":
    arrs, nw, nr = bench.make_windows("C5", 2048, 10_000, 8, 16)
    p.num_samples = 3
else:
    arrs, nw, nr = bench.make_windows("C2", 1024, 10_000, 8, 16)
arrs, nw, nr = synth.tile_batch(arrs, nw, nr, max(1, n // nw))
dev = torch.device("cuda", 0)
d = {k: torch.from_numpy(v.view(np.uint8) if v.dtype != np.uint8 else v).to(dev) for k, v in arrs.items()}
b = capi.make_batch_struct(d, nw, nr)


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
    print("%s lanes %d: %.1f ms per step of %d windows; kernels summed %.1f ms:" % (leg, lanes, dt * 1e3, nw, sum(kt.values())),
          ", ".join("%s %.1f" % (k, x) for k, x in sorted(kt.items(), key=lambda y: -y[1])[:16]), flush=True)
    print("   stats:", eng.stats(), flush=True)
    eng.close()
