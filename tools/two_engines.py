"""Experiment: two engines (two HIP streams, two host threads) each processing half of the batch concurrently."""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lancet2_amd import capi, synth  # noqa: E402
from lancet2_amd.engine import Engine  # noqa: E402

NE = int(sys.argv[1]) if len(sys.argv) > 1 else 2
NW = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
os.environ["MA_HBM_SHARE"] = str(1.0 / NE)
dev = torch.device("cuda", 0)
params = capi.default_params(min_k=25, max_k=25)
engines = []
for e in range(NE):
    arrs, n0, nr0 = synth.make_config_batch("C2", 64, first_index=10_000 + 64 * e)
    arrs, n, nr = synth.tile_batch(arrs, n0, nr0, NW // NE // 64)
    d = {k: torch.from_numpy(v.view(np.uint8) if v.dtype != np.uint8 else v).to(dev) for k, v in arrs.items()}
    b = capi.make_batch_struct(d, n, nr)

    def alloc(spec):
        return {k: torch.zeros(int(sz) * np.dtype(dt).itemsize, dtype=torch.uint8, device=dev) for k, (dt, sz) in spec.items()}
    bufs = [alloc(capi.gate_out_spec(n)), alloc(capi.asm_out_spec(params, n)), alloc(capi.var_out_spec(params, n)),
            alloc(capi.geno_out_spec(params, n, nr, debug=False))]
    structs = [capi.fill_struct(c, x) for c, x in zip((capi.GateOut, capi.AsmOut, capi.VarOut, capi.GenoOut), bufs)]
    eng = Engine(params, device=0, memspace=capi.MA_MEM_DEVICE)
    st = torch.cuda.Stream(dev)
    eng.set_stream(st.cuda_stream)
    eng.timing_control(0)
    engines.append((eng, b, structs, d, bufs, st, n))


def run(i):
    eng, b, structs = engines[i][:3]
    eng.process_device(b, *structs)


def step():
    ts = [threading.Thread(target=run, args=(i,)) for i in range(NE)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize(dev)


step()
t0 = time.perf_counter()
K = 3
for _ in range(K):
    step()
el = time.perf_counter() - t0
tot = sum(e[6] for e in engines)
print(f"engines {NE}: {tot * K / el:.0f} windows/s, {el / K * 1e3:.1f} ms per step of {tot} windows")
