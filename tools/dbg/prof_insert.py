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
buf = (C.c_ulonglong * 32)()
eng.lib.ma_debug_iprof(buf)
base = list(buf)
eng.timing_control(1)
os.environ["MA_VERBOSE"] = "1"
eng.process(arrs, nw, nr)
os.environ.pop("MA_VERBOSE")
eng.lib.ma_debug_iprof(buf)
d = [b - a for a, b in zip(base, list(buf))]
names = ["staging", "ids (ref + slow)", "map: ref k-mers", "map: slow queue", "table out", "general route", "-"]
tot = sum(d[:7])
print({k: round(v, 2) for k, v in eng.kernel_times()})
print({n: f"{100.0 * v / tot:.1f}%" for n, v in zip(names, d)})
cn = ["stage", "lane loop", "slow queue", "-"]
ct = sum(d[8:12]) or 1
mt = sum(d[12:16]) or 1
print("k_mm_lds", {n: f"{100.0 * v / mt:.1f}%" for n, v in zip(["init+bases+leaders", "scan+queue+inserts", "compaction", "support"], d[12:16])})
print("k_classify", {n: f"{100.0 * v / ct:.1f}%" for n, v in zip(cn, d[8:12])}, "mean cycles(100MHz ticks) per tile:", ct / max(1, 2048 * 11))
gt = sum(d[16:22]) or 1
print("k_graph", {n: f"{100.0 * v / gt:.1f}%" for n, v in zip(["table pass", "ranks", "node records", "ref nodes + set init", "edge pass", "edge lists"], d[16:22])})
qt = sum(d[22:26]) or 1
print("k_mm_q", {n: f"{100.0 * v / qt:.1f}%" for n, v in zip(["set init", "queue -> set", "compaction", "support"], d[22:26])})
eng.close()
print("k_support (after its set-up)", {n: round(v / 2048.0) for n, v in zip(["group loop", "flush"], d[27:29])})
print("raw ticks per workgroup (100 MHz?):", {i: round(v / 2048.0, 1) for i, v in enumerate(d) if v})
