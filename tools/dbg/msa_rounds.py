"""Developer tool: the POA rounds of the bench workload -- how many windows still have an alignment pending in round r,
and what each k_msa / k_msa_band launch costs (single lane, HIP events).  usage: python tools/dbg/msa_rounds.py [n]"""
import argparse
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import bench  # noqa: E402
from lancet2_amd import capi  # noqa: E402
from lancet2_amd.engine import Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("n", nargs="?", type=int, default=2048)
ap.add_argument("--str-every", type=int, default=8)
ap.add_argument("--cascade", action="store_true", help="the reference default ladder k = 13 ... 127 instead of k = 25")
a = ap.parse_args()
arrs, n, nr = bench.make_windows("C3", a.n, 10_000, a.str_every, 8)
eng = Engine(capi.default_params() if a.cascade else capi.default_params(min_k=25, max_k=25))
eng.set_streams(1)
g, asm, v, q = eng.process(arrs, n, nr, debug=False)
eng.timing_control(1)
g, asm, v, q = eng.process(arrs, n, nr, debug=False)
nh = asm["comp_nhaps"].reshape(n, -1).astype(np.int64)
nc = asm["win_ncomp"]
al = np.array([sum(max(int(nh[w, c]) - 1, 0) for c in range(int(nc[w]))) for w in range(n)])
print("alignments per window histogram:", np.bincount(al).tolist())
print("haplotype length: mean %.0f max %d" % (asm["hap_len"][asm["hap_len"] > 0].mean(), asm["hap_len"].max()))
for name, ms in eng.kernel_times():
    if name.startswith("k_msa"):
        print(f"  {name:12s} {ms:7.3f} ms")
agg = {}
for k, t in eng.kernel_times():
    agg[k] = agg.get(k, 0.0) + t
print({k: round(t, 2) for k, t in sorted(agg.items(), key=lambda kv: -kv[1])})
eng.close()
