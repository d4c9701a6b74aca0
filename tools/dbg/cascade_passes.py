"""Developer tool: the reference's default k ladder on one lane -- per kernel: launches, total and mean time."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from lancet2_amd import capi
from lancet2_amd import engine as E
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
arrs, nw, nr = bench.make_windows("C3", n, 10_000, 8, 8)
eng = E.Engine(capi.default_params(min_k=13, max_k=127, k_step=6))
eng.set_streams(1)
eng.process(arrs, nw, nr)
eng.timing_control(1)
t = time.perf_counter()
eng.process(arrs, nw, nr)
dt = time.perf_counter() - t
agg = collections.OrderedDict()
for k, v in eng.kernel_times():
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += v
print("wall %.1f ms (host arrays: includes PCIe); kernels %.1f ms" % (dt * 1e3, sum(v[1] for v in agg.values())))
for k, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-16s launches %3d  total %7.2f ms  mean %6.3f ms" % (k, c, ms, ms / c))
print(eng.stats())
eng.close()
