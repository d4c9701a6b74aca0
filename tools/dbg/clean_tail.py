"""Developer tool: distribution of k_clean's per-window time on the bench workload (-DMA_PROFILE build): the kernel lasts as
long as its slowest window."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from lancet2_amd import capi
from lancet2_amd import engine as E
capi.LIB_PATH = os.path.join(capi.REPO, "lancet2_amd", "libmicroasm_prof.so")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
arrs, nw, nr = bench.make_windows("C3", n, 10_000, 8, 8)
eng = E.Engine(capi.default_params(min_k=25, max_k=25))
eng.set_streams(1)
eng.process(arrs, nw, nr)
eng.timing_control(1)
eng.process(arrs, nw, nr)
print({k: round(v, 2) for k, v in eng.kernel_times() if k == "k_clean"})
buf = (C.c_ulonglong * (4 * n))()
eng.lib.ma_debug_cwin(buf, n)
a = np.array(list(buf), dtype=np.uint64).reshape(n, 4)
t = a[:, 0].astype(np.float64)
ok = t > 0
print("windows with a record:", int(ok.sum()), "of", n)
q = np.quantile(t[ok], [0.5, 0.9, 0.99, 1.0])
print("ticks: median %.0f  p90 %.0f  p99 %.0f  max %.0f  mean %.0f" % (q[0], q[1], q[2], q[3], t[ok].mean()))
tb = (C.c_ulonglong * (6 * n))()
eng.lib.ma_debug_ctime(tb, n)
ph = np.array(list(tb), dtype=np.float64).reshape(n, 6)
order = np.argsort(-t)
names = ["comps+anchors", "compress+tips", "index+cycle+cx", "maxflow", "emit", "?"]
for i in order[:10]:
    print("win %4d ticks %9.0f nodes %5d  " % (i, t[i], a[i, 1]) + "  ".join("%s %.0f%%" % (names[x], 100 * ph[i, x] / max(t[i], 1)) for x in range(5)))
tot = ph[ok].sum(axis=0)
print("all windows: " + "  ".join("%s %.0f%%" % (names[x], 100 * tot[x] / t[ok].sum()) for x in range(5)))
eng.close()
