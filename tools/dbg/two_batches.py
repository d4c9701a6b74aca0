"""Developer experiment: TWO batches in flight -- two engines (own workspaces, own lanes), each stepping its own 16384-window
batch from its own host thread -- against one engine stepping twice as often: does the next batch's build stage fill the
tail of the previous batch's genotype stage?  (usage: python3 tools/dbg/two_batches.py [steps]; MA_BENCH_CACHE as for bench.py)"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from lancet2_amd import capi, synth  # noqa: E402
from lancet2_amd.engine import Engine  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
arrs0, n0, nr0 = bench.make_windows("C3", 16384, 10_000, 8, 1)
arrs, n, nr = synth.tile_batch(arrs0, n0, nr0, 1)
params = capi.default_params(min_k=25, max_k=25)


def setup():
    d = {k: torch.from_numpy(v.view(np.uint8) if v.dtype != np.uint8 else v).to(dev) for k, v in arrs.items()}
    b = capi.make_batch_struct(d, n, nr)

    def alloc(spec):
        return {k: torch.zeros(int(sz) * np.dtype(dt).itemsize, dtype=torch.uint8, device=dev) for k, (dt, sz) in spec.items()}
    bufs = [alloc(capi.gate_out_spec(n)), alloc(capi.asm_out_spec(params, n)), alloc(capi.var_out_spec(params, n)),
            alloc(capi.geno_out_spec(params, n, nr, debug=False))]
    st = [capi.fill_struct(c, x) for c, x in zip((capi.GateOut, capi.AsmOut, capi.VarOut, capi.GenoOut), bufs)]
    eng = Engine(params, device=0, memspace=capi.MA_MEM_DEVICE)
    return eng, b, st, (d, bufs)


def run(engs, steps_each):
    for e, b, st, _ in engs:
        e.process_device(b, *st)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()

    def work(x):
        e, b, st, _ = engs[x]
        for _ in range(steps_each):
            e.process_device(b, *st)
    th = [threading.Thread(target=work, args=(x,)) for x in range(len(engs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    total = steps_each * len(engs)
    return dt * 1e3 / total


os.environ["MA_HBM_SHARE"] = os.environ.get("MA_HBM_SHARE", "0.5")
engs = [setup(), setup()]
one = run(engs[:1], 2 * K)
two = run(engs, K)
one2 = run(engs[:1], 2 * K)
print(f"one batch in flight: {one:.2f} / {one2:.2f} ms per 16384-window step; two in flight: {two:.2f} ms per step  ({one / two:.3f}x)")
