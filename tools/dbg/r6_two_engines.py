"""Developer experiment (round 6, one chunk per lane): do the lanes lose to the barrier at the end of every batch?  ONE engine
with four lanes stepping a 16384-window batch (all lanes join after every step) against TWO engines with two lanes each, each
stepping its own 8192-window batch from its own host thread, never waiting for the other -- the same four lanes of 4096
windows, without the common barrier.  (usage: python3 tools/dbg/r6_two_engines.py [steps]; MA_BENCH_CACHE as for bench.py)"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from lancet2_amd import capi  # noqa: E402
from lancet2_amd.engine import Engine  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
params = capi.default_params(min_k=25, max_k=25)
full = bench.make_windows("C3", 16384, 10_000, 8, 1)
halves = [bench.make_windows("C3", 8192, 10_000, 8, 1), bench.make_windows("C3", 8192, 10_000 + 8192, 8, 1)]


def setup(batch, lanes, share):
    arrs, n, nr = batch
    d = {k: torch.from_numpy(v.view(np.uint8) if v.dtype != np.uint8 else v).to(dev) for k, v in arrs.items()}
    b = capi.make_batch_struct(d, n, nr)

    def alloc(spec):
        return {k: torch.zeros(int(sz) * np.dtype(dt).itemsize, dtype=torch.uint8, device=dev) for k, (dt, sz) in spec.items()}
    bufs = [alloc(capi.gate_out_spec(n)), alloc(capi.asm_out_spec(params, n)), alloc(capi.var_out_spec(params, n)),
            alloc(capi.geno_out_spec(params, n, nr, debug=False))]
    st = [capi.fill_struct(c, x) for c, x in zip((capi.GateOut, capi.AsmOut, capi.VarOut, capi.GenoOut), bufs)]
    os.environ["MA_HBM_SHARE"] = str(share)
    eng = Engine(params, device=0, memspace=capi.MA_MEM_DEVICE)
    eng.set_streams(lanes)
    return eng, b, st, (d, bufs), n


def run(engs, steps_each):
    for e, b, st, _, _ in engs:
        for _ in range(2):
            e.process_device(b, *st)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()

    def work(x):
        e, b, st, _, _ = engs[x]
        for _ in range(steps_each):
            e.process_device(b, *st)
    th = [threading.Thread(target=work, args=(x,)) for x in range(len(engs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return sum(e[4] for e in engs) * steps_each / dt


mode = os.environ.get("MODE", "both")
if mode in ("one", "both"):
    one = [setup(full, 4, 1.0)]
    print("one engine, four lanes, 16384 windows per step: %.1f k submitted windows/s" % (run(one, K) / 1e3), flush=True)
    one[0][0].close()
    del one
    torch.cuda.empty_cache()
if mode in ("two", "both"):
    two = [setup(halves[0], 2, 0.5), setup(halves[1], 2, 0.5)]
    print("two engines, two lanes each, 8192 windows per step each, no common barrier: %.1f k submitted windows/s" % (run(two, 2 * K) / 1e3), flush=True)
