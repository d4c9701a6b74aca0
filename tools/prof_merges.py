"""Developer tool: per-window chain-merge statistics of k_clean from a -DMA_PROFILE build (see tools/prof_phases.py)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lancet2_amd import capi, synth  # noqa: E402
from lancet2_amd import engine as E  # noqa: E402

capi.LIB_PATH = os.path.join(capi.REPO, "lancet2_amd", "libmicroasm_prof.so")
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
arrs, nw, nr = synth.make_config_batch(cfg, 64)
if len(sys.argv) > 2:
    arrs, nw, nr = synth.tile_batch(arrs, nw, nr, int(sys.argv[2]) // 64)
eng = E.Engine(capi.default_params(min_k=25, max_k=25))
eng.process(arrs, nw, nr)
buf = (C.c_uint * (8 * 64))()
eng.lib.ma_debug_cmerge(buf, 64)
m = np.array(list(buf), dtype=np.int64).reshape(64, 8)
names = ["merges_ph0", "merges_later", "maxwalk_ph0", "maxwalk_later", "walks_ph0", "walks_later", "nodes"]
for j, nm in enumerate(names):
    print("%-14s mean %8.1f  median %6d  max %6d" % (nm, m[:, j].mean(), np.median(m[:, j]), m[:, j].max()))
print(m[:8, :7])
tb = (C.c_ulonglong * (6 * 64))()
eng.lib.ma_debug_ctime(tb, 64)
t = np.array(list(tb), dtype=np.int64).reshape(64, 6)
for j, nm in enumerate(['filter', 'hop', 'validate', 'apply', 'compress_graph_total', 'init+links']):
    print('%-22s mean ticks %10.0f' % (nm, t[:, j].mean()))
eng.close()
