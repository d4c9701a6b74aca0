import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from lancet2_amd import capi, synth
from lancet2_amd import engine as E
capi.LIB_PATH = os.path.join(capi.REPO, "lancet2_amd", "libmicroasm_prof.so")
arrs, nw, nr = synth.make_config_batch("C3", 64, first_index=10000)
arrs, nw, nr = synth.tile_batch(arrs, nw, nr, 32)
eng = E.Engine(capi.default_params(min_k=25, max_k=25))
eng.set_streams(1)
eng.process(arrs, nw, nr)
buf = (C.c_ulonglong * 16)()
eng.lib.ma_debug_iprof(buf)
base = list(buf)
eng.timing_control(1)
eng.process(arrs, nw, nr)
eng.lib.ma_debug_iprof(buf)
d = [b - a for a, b in zip(base, list(buf))]
names = ["stage", "unused", "setup+walk", "queue"]
tot = sum(d[:4])
print({k: round(v, 2) for k, v in eng.kernel_times() if k in ("k_classify",)})
print({n: f"{100.0 * v / tot:.1f}%" for n, v in zip(names, d)})
eng.close()
