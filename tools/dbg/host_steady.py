"""Developer tool: steady state of the host route (MA_MEM_HOST, ma_prefetch_batch) next to the device-resident route on the
same batch.  usage: host_steady.py [host|device|both] [n_windows] [steps]; MA_VERBOSE_LAST=1 prints the lanes' phase times
of the last step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
import bench
from lancet2_amd import capi, synth
from lancet2_amd.engine import Engine
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
distinct = min(n, int(os.environ.get("DISTINCT", "2048")))
arrs, n0, nr0 = bench.make_windows("C3", distinct, 10_000, 8, int(os.environ.get("WORKERS", "8")))  # WORKERS=1 under rocprofv3 (forked pool workers hang in its signal handler)
arrs, n, nr = synth.tile_batch(arrs, n0, nr0, n // distinct)
params = capi.default_params(min_k=25, max_k=25)
keep = []
def pinned(a):
    t = torch.empty(max(a.nbytes, 16), dtype=torch.uint8, pin_memory=True); keep.append(t)
    v = t.numpy()[:a.nbytes].view(a.dtype); v[...] = a; return v
def start_bg_copier(dev):
    """BG_COPY=<chunk MB>: a background thread streams pinned host memory to the device on a side stream meanwhile
    (interference experiment: what does a busy PCIe link cost the kernels and their host round trips?)"""
    stop = []
    moved = [0]
    if not os.environ.get("BG_COPY"):
        return None, stop, moved
    import threading
    chunk = int(float(os.environ["BG_COPY"]) * (1 << 20))
    gap_us = float(os.environ.get("BG_GAP_US", "0"))
    span = max(chunk, int(float(os.environ.get("BG_SPAN_MB", "0")) * (1 << 20))) // chunk * chunk
    burst = int(os.environ.get("BG_BURST", "1"))  # pieces queued back to back before the thread waits for them
    src = torch.empty(span, dtype=torch.uint8, pin_memory=True)
    dst = torch.empty(span, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    def bg():
        o = 0
        with torch.cuda.stream(side):
            while not stop:
                for _ in range(burst):
                    dst[o:o + chunk].copy_(src[o:o + chunk], non_blocking=True)
                    o = (o + chunk) % span
                side.synchronize()
                moved[0] += chunk * burst
                if gap_us:
                    time.sleep(gap_us * 1e-6)
    th = threading.Thread(target=bg); th.start()
    return th, stop, moved
def run_host():
    h_in = {k: pinned(np.ascontiguousarray(v)) for k, v in arrs.items()}
    def pout(spec):
        return {k: pinned(np.zeros(int(sz), dtype=dt)) for k, (dt, sz) in spec.items()}
    outs = (pout(capi.gate_out_spec(n)), pout(capi.asm_out_spec(params, n)), pout(capi.var_out_spec(params, n)), pout(capi.geno_out_spec(params, n, nr, debug=False)))
    st = (capi.fill_struct(capi.GateOut, outs[0]), capi.fill_struct(capi.AsmOut, outs[1]), capi.fill_struct(capi.VarOut, outs[2]), capi.fill_struct(capi.GenoOut, outs[3]))
    b = capi.make_batch_struct(h_in, n, nr)
    eng = Engine(params, memspace=capi.MA_MEM_HOST)
    eng.timing_control(int(os.environ.get("TIMING", "0")))
    for _ in range(2):
        eng.prefetch(b); eng.process_device(b, *st)
    if mode == "upload":  # the upload alone, nothing computing: MA_VERBOSE prints its rate
        os.environ["MA_VERBOSE"] = "1"
        for _ in range(3):
            eng.prefetch(b); time.sleep(0.2); eng.process_device(b, *st)
        os.environ.pop("MA_VERBOSE")
        eng.close()
        return
    eng.prefetch(b)
    eng.prefetch(b); eng.process_device(b, *st)   # one pipelined step before the clock starts: the steady state is what is timed
    th, stop, moved = start_bg_copier(torch.device("cuda:0"))
    t = time.perf_counter()
    tp = 0.0
    for it in range(steps):
        t1 = time.perf_counter()
        eng.prefetch(b)
        tp += time.perf_counter() - t1
        if it == steps - 1 and os.environ.get("MA_VERBOSE_LAST"):
            os.environ["MA_VERBOSE"] = "1"
        eng.process_device(b, *st)
    dt = (time.perf_counter() - t) / steps
    print("prefetch call: %.2f ms of host time per step" % (tp / steps * 1e3))
    os.environ.pop("MA_VERBOSE", None)
    if th:
        stop.append(1); th.join()
        print("background H2D: %.1f GB/s in chunks of %s MB" % (moved[0] / (dt * steps) / 1e9, os.environ["BG_COPY"]))
    print("host route:", round(dt * 1e3, 2), "ms/step ->", round(n / dt), "windows/s", flush=True)
    eng.close()
    return dt
def run_device():
    dev = torch.device("cuda:0")
    d_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in arrs.items()}
    def dout(spec):
        return {k: torch.zeros(int(sz) * np.dtype(dt).itemsize, dtype=torch.uint8, device=dev) for k, (dt, sz) in spec.items()}
    outs = (dout(capi.gate_out_spec(n)), dout(capi.asm_out_spec(params, n)), dout(capi.var_out_spec(params, n)), dout(capi.geno_out_spec(params, n, nr, debug=False)))
    st = (capi.fill_struct(capi.GateOut, outs[0]), capi.fill_struct(capi.AsmOut, outs[1]), capi.fill_struct(capi.VarOut, outs[2]), capi.fill_struct(capi.GenoOut, outs[3]))
    b = capi.make_batch_struct(d_in, n, nr)
    eng = Engine(params, memspace=capi.MA_MEM_DEVICE)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.timing_control(int(os.environ.get("TIMING", "0")))
    for _ in range(2):
        eng.process_device(b, *st)
    torch.cuda.synchronize()
    th, stop, moved = start_bg_copier(dev)
    t = time.perf_counter()
    for it in range(steps):
        eng.process_device(b, *st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    if th:
        stop.append(1); th.join()
        print("background H2D: %.1f GB/s in chunks of %s MB" % (moved[0] / (dt * steps) / 1e9, os.environ["BG_COPY"]))
    print("device route:", round(dt * 1e3, 2), "ms/step ->", round(n / dt), "windows/s", flush=True)
    eng.close()
if mode in ("device", "both"):
    run_device()
if mode in ("host", "both", "upload"):
    run_host()
