"""Developer tool: which of the bench's deep-panel windows carry which flag (first pass only with MA_NO_CAP_RETRY=1)."""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import bench  # noqa: E402
from lancet2_amd import capi  # noqa: E402
from lancet2_amd.engine import Engine  # noqa: E402

nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 512
bench.NOHINT_EVERY = 50
arrs, n, nr = bench.make_windows("C4", nwin, 10_000, 0, 16)
params = capi.default_params(min_k=25, max_k=25)
for env in ({}, {"MA_NO_CAP_RETRY": "1"}):
    os.environ.pop("MA_NO_CAP_RETRY", None)
    os.environ.update(env)
    eng = Engine(params)
    g, a, v, q = eng.process(arrs, n, nr, debug=False)
    eng.close()
    st = a["win_status"]
    print("env", env)
    for nm in ("MA_W_NO_HAPLOTYPE", "MA_W_HAP_OVERFLOW", "MA_W_LEN_OVERFLOW", "MA_W_BFS_LIMIT", "MA_W_TABLE_OVERFLOW", "MA_W_VAR_OVERFLOW",
               "MA_W_CIGAR_OVERFLOW"):
        idx = np.nonzero(st & getattr(capi, nm))[0]
        print(" ", nm, len(idx), (idx[:40] + 10_000).tolist())
    ncomp = a["win_ncomp"]
    nh = a["comp_nhaps"].reshape(n, params.max_comps).sum(axis=1)
    print("  max comps", int(ncomp.max()), "max haps", int(nh.max()))
