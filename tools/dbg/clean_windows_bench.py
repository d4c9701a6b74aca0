"""Developer tool: distribution of per-window k_clean time on the bench workload (a -DMA_PROFILE build, see
tools/prof_phases.py).  usage: python tools/dbg/clean_windows_bench.py [n<=4096]"""
import ctypes as C
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import bench  # noqa: E402
from lancet2_amd import capi  # noqa: E402
from lancet2_amd import engine as E  # noqa: E402

capi.LIB_PATH = os.path.join(capi.REPO, "lancet2_amd", os.environ.get("MA_PROF_LIB", "libmicroasm_prof.so"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
arrs, nw, nr = bench.make_windows("C3", n, 10_000, 8, 8)
eng = E.Engine(capi.default_params(min_k=25, max_k=25))
eng.set_streams(1)
eng.process(arrs, nw, nr)
eng.process(arrs, nw, nr)
buf = (C.c_ulonglong * (4 * n))()
eng.lib.ma_debug_cwin(buf, n)
tb = (C.c_ulonglong * (6 * n))()
eng.lib.ma_debug_ctime(tb, n)
t = np.array([buf[4 * i] for i in range(n)], dtype=np.float64)
nodes = np.array([buf[4 * i + 1] for i in range(n)])
print("windows", n, "ticks: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % (t.mean(), *np.percentile(t, [50, 90, 99]), t.max()))
order = np.argsort(-t)
for i in list(order[:12]) + list(order[n // 2:n // 2 + 3]):
    print("win %4d (str %d) ticks %9d nodes %5d rounds %3d cands %2d | tips %8d index %8d cyc+cx %8d maxflow %8d emit %8d haps %3d" %
          ((i, int((10_000 + i) % 8 == 7), int(t[i]), nodes[i], buf[4 * i + 2], buf[4 * i + 3]) + tuple(tb[6 * i + q] for q in range(6))))
eng.close()
