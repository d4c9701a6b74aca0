import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from lancet2_amd import capi, synth
from lancet2_amd import engine as E
capi.LIB_PATH = os.path.join(capi.REPO, "lancet2_amd", "libmicroasm_prof.so")
import bench
arrs, nw, nr = bench.make_windows("C3", 2048, 10_000, 8, 8)
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
names = ["init+stage", "ref k-mers", "slow pass 1", "table init", "pass 2 inserts", "deferred", "pass 3"]
tot = sum(d[:7])
print({k: round(v, 2) for k, v in eng.kernel_times()})
print({n: f"{100.0 * v / tot:.1f}%" for n, v in zip(names, d)})
cn = ["stage", "lane loop", "slow queue", "-"]
ct = sum(d[8:12]) or 1
mt = sum(d[12:16]) or 1
print("k_mm_lds", {n: f"{100.0 * v / mt:.1f}%" for n, v in zip(["init+bases+leaders", "scan+queue+inserts", "compaction", "support"], d[12:16])})
print("k_classify", {n: f"{100.0 * v / ct:.1f}%" for n, v in zip(cn, d[8:12])}, "mean cycles(100MHz ticks) per tile:", ct / max(1, 2048 * 11))
eng.close()
