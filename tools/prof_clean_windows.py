"""Developer tool: per-window k_clean cycles from a -DMA_PROFILE build (see tools/prof_phases.py)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lancet2_amd import capi, synth  # noqa: E402
from lancet2_amd import engine as E  # noqa: E402

capi.LIB_PATH = os.path.join(capi.REPO, "lancet2_amd", "libmicroasm_prof.so")
arrs, nw, nr = synth.make_config_batch(sys.argv[2] if len(sys.argv) > 2 else "C2", 64, first_index=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
eng = E.Engine(capi.default_params(min_k=25, max_k=25))
eng.process(arrs, nw, nr)
eng.process(arrs, nw, nr)
buf = (C.c_ulonglong * (4 * 64))()
eng.lib.ma_debug_cwin(buf, 64)
rows = [(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3], i) for i in range(64)]
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("mean ticks", tot // 64, "max", rows[0][0])
for r in rows[:12]:
    print("win %2d ticks %10d nodes %5d comps %4d cands %2d" % (r[4], r[0], r[1], r[2], r[3]))
tb = (C.c_ulonglong * (6 * 64))()
eng.lib.ma_debug_ctime(tb, 64)
for r in rows[:4] + rows[30:33] + rows[-2:]:
    i = r[4]
    print("win %2d total %8d  comps+anchors %8d  compress+tips %8d  index+cycle+cx %8d  maxflow %8d  emit %8d" % ((i, r[0]) + tuple(tb[6 * i + q] for q in range(5))))
print("...")
for r in rows[-4:]:
    print("win %2d ticks %10d nodes %5d comps %4d cands %2d" % (r[4], r[0], r[1], r[2], r[3]))
eng.close()
