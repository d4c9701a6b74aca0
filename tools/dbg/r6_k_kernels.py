"""Developer tool: single-lane kernel times of ONE k (min_k = max_k = k) on the bench workload, for the rungs of the ladder.
usage: python tools/dbg/r6_k_kernels.py [windows] [k ...]"""
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
ks = [int(x) for x in sys.argv[2:]] or [13, 19, 25]
arrs, nw, nr = bench.make_windows("C3", n, 10_000, 8, 16)
dev = torch.device("cuda", 0)
d = {k: torch.from_numpy(v.view(np.uint8) if v.dtype != np.uint8 else v).to(dev) for k, v in arrs.items()}
b = capi.make_batch_struct(d, nw, nr)
for kk in ks:
    p = capi.default_params(min_k=kk, max_k=kk)

    def alloc(spec):
        return {k: torch.zeros(int(sz) * np.dtype(dt).itemsize, dtype=torch.uint8, device=dev) for k, (dt, sz) in spec.items()}
    g, a, v, q = alloc(capi.gate_out_spec(nw)), alloc(capi.asm_out_spec(p, nw)), alloc(capi.var_out_spec(p, nw)), alloc(capi.geno_out_spec(p, nw, nr, debug=False))
    st = (capi.fill_struct(capi.GateOut, g), capi.fill_struct(capi.AsmOut, a), capi.fill_struct(capi.VarOut, v), capi.fill_struct(capi.GenoOut, q))
    eng = Engine(p, device=0, memspace=capi.MA_MEM_DEVICE)
    eng.set_streams(1)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    eng.timing_control(0)
    eng.process_device(b, *st)
    torch.cuda.synchronize()
    eng.timing_control(3)
    eng.process_device(b, *st)
    torch.cuda.synchronize()
    stats = eng.stats()
    eng.timing_control(2)
    t = time.perf_counter()
    for _ in range(2):
        eng.process_device(b, *st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 2
    kt = {}
    for name, ms in eng.kernel_times():
        kt[name] = kt.get(name, 0.0) + ms / 2
    status = a["win_status"].view(torch.int32).cpu().numpy().view(np.uint32)
    ok = int(((status & capi.MA_W_NO_HAPLOTYPE) == 0).sum())
    print("k %d: %.1f ms per step, %d of %d windows assembled; kernels summed %.1f ms:" % (kk, dt * 1e3, ok, nw, sum(kt.values())),
          ", ".join("%s %.1f" % (k, x) for k, x in sorted(kt.items(), key=lambda y: -y[1])[:16]))
    print("    per window: slow instances %.0f, distinct k-mers %.0f, nodes after low-cov %.0f, edge queue %.0f, count queue %.0f; attempts %d" % tuple(
        [stats.get(x, 0) / max(stats.get("window_attempts", 1), 1) for x in ("slow_instances", "distinct_kmers", "nodes_after_lowcov", "edge_queue", "count_queue")] + [stats.get("window_attempts", 0)]), flush=True)
    eng.close()
