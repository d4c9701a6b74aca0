"""Developer tool: kernel times of the clean stage (k_clean_chains / k_clean_tail / k_clean) on the bench workload, single lane.
usage: python tools/dbg/clean_split.py [n_windows]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import bench  # noqa: E402
from lancet2_amd import capi  # noqa: E402
from lancet2_amd import engine as E  # noqa: E402

if os.environ.get("MA_LIB"):
    capi.LIB_PATH = os.path.join(capi.REPO, "lancet2_amd", os.environ["MA_LIB"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cfg = sys.argv[2] if len(sys.argv) > 2 else "C3"
arrs, nw, nr = bench.make_windows(cfg, n, 10_000, 8 if cfg != "C4" else 0, 8)
eng = E.Engine(capi.default_params(min_k=25, max_k=25))
eng.set_streams(1)
eng.process(arrs, nw, nr)
eng.timing_control(1)
eng.process(arrs, nw, nr)
acc = {}
for k, v in eng.kernel_times():
    acc[k] = acc.get(k, 0.0) + v
print({k: round(v, 3) for k, v in acc.items()})
print("clean launches:", [(k, round(v, 3)) for k, v in eng.kernel_times() if "clean" in k])
eng.close()
