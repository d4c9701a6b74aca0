"""Developer tool: step time (device-resident, default lanes) of the k = 25 headline and of the default ladder for several values of
MA_POA_MIN_PENDING (how many windows must wait for a band fill before the host launches another band round).
usage: MA_POA_MIN_PENDING=<n> python tools/dbg/poa_pending_ab.py [windows]"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench  # noqa: E402
from lancet2_amd import capi  # noqa: E402
from lancet2_amd import engine as E  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
arrs, nw, nr = bench.make_windows("C3", n, 10_000, 8, 8)
for tag, p in (("k25", capi.default_params(min_k=25, max_k=25)), ("ladder", capi.default_params())):
    eng = E.Engine(p)
    eng.process(arrs, nw, nr)
    eng.timing_control(0)
    t = time.perf_counter()
    for _ in range(3):
        eng.process(arrs, nw, nr)
    dt = (time.perf_counter() - t) / 3
    print(os.environ.get("MA_POA_MIN_PENDING", "default"), tag, "%.1f ms per step (host arrays: incl. PCIe)" % (dt * 1e3))
    eng.close()
