"""Developer tool: per-phase cycle breakdown of k_vote / k_msa from a -DMA_PROFILE build.

  make -C lancet2_amd/csrc clean && make -C lancet2_amd/csrc HIPFLAGS_EXTRA=-DMA_PROFILE LIB=../libmicroasm_prof.so
  python tools/prof_phases.py [n_windows [bench]]

Not part of the product path or the tests.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lancet2_amd import capi, synth  # noqa: E402
from lancet2_amd import engine as E  # noqa: E402

capi.LIB_PATH = os.path.join(capi.REPO, "lancet2_amd", os.environ.get("MA_PROF_LIB", "libmicroasm_prof.so"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
if len(sys.argv) > 2 and sys.argv[2] == "bench":  # the bench workload: n distinct windows, every 8th with a tandem repeat
    import bench
    arrs, nw, nr = bench.make_windows("C3", n, 10_000, 8, 8)
else:
    arrs, nw, nr = synth.make_config_batch("C3", 64)
    arrs, nw, nr = synth.tile_batch(arrs, nw, nr, n // 64)
eng = E.Engine(capi.default_params(min_k=25, max_k=25))
eng.set_streams(1)
eng.process(arrs, nw, nr)
buf = (C.c_ulonglong * 16)()
for sym, names in (("ma_debug_chprof", ["load", "fastsv", "number", "anchors+cands", "classify", "states+jump", "positions", "segtable", "values", "simulate", "compact", "records", "strings", "sim_scan", "sim_turn", "x14", "x15"]), ("ma_debug_cprof", ["init", "components", "anchors", "compress1", "lowcov", "compress2", "tips", "trav_index", "cycle+cx", "maxflow", "emit", "e_rank+tables", "e_refhap", "e_buildseq", "e_dedup", "e_finalize+misc"]), ("ma_debug_vprof", ["encode", "vote", "argmax", "second", "clear", "mismatch", "emit", "pairs", "shortcut", "shortcut_hits", "shortcut_miss", "x11", "x12", "x13", "wave_total", "wg_setup(thread 0)"]),
                   ("ma_debug_prof2", ["sched", "G_jobs", "F_phases", "idle", "F1", "F2", "F3", "F4", "n_G", "n_idle_polls", "wg_total"]),
                   ("ma_debug_prof", ["prep", "fill", "end+traceback", "add_alignment", "toposort", "extract", "first_seq", "img_restore", "fill_pre", "fill_barrier", "fill_post", "rows_generic", "rows", "band_fallbacks", "img_save"])):
    fn = getattr(eng.lib, sym, None)
    if fn is None:
        continue
    fn(buf, 1)
eng.timing_control(1)
eng.process(arrs, nw, nr)
print({k: round(v, 2) for k, v in eng.kernel_times()})
for sym, names in (("ma_debug_chprof", ["load", "fastsv", "number", "anchors+cands", "classify", "states+jump", "positions", "segtable", "values", "simulate", "compact", "records", "strings", "sim_scan", "sim_turn", "x14", "x15"]), ("ma_debug_cprof", ["init", "components", "anchors", "compress1", "lowcov", "compress2", "tips", "trav_index", "cycle+cx", "maxflow", "emit", "e_rank+tables", "e_refhap", "e_buildseq", "e_dedup", "e_finalize+misc"]), ("ma_debug_vprof", ["encode", "vote", "argmax", "second", "clear", "mismatch", "emit", "pairs", "shortcut", "shortcut_hits", "shortcut_miss", "x11", "x12", "x13", "wave_total", "wg_setup(thread 0)"]),
                   ("ma_debug_prof2", ["sched", "G_jobs", "F_phases", "idle", "F1", "F2", "F3", "F4", "n_G", "n_idle_polls", "wg_total"]),
                   ("ma_debug_prof", ["prep", "fill", "end+traceback", "add_alignment", "toposort", "extract", "first_seq", "img_restore", "fill_pre", "fill_barrier", "fill_post", "rows_generic", "rows", "band_fallbacks", "img_save"])):
    fn = getattr(eng.lib, sym, None)
    if fn is None:
        continue
    fn(buf, 0)
    vals = list(buf)[:len(names)]
    print(sym, {k: v for k, v in zip(names, vals)})
fn = getattr(eng.lib, "ma_debug_vhist", None)
if fn is not None:  # k_vote: how long its workgroups live (buckets of 131072 clock ticks)
    hb = (C.c_ulonglong * 40)()
    fn(hb, 0)
    h = list(hb)
    print("k_vote workgroup lifetimes, buckets of 131072 ticks:", h[:32], "longest", h[32] >> 24, "item", h[32] & 0xFFFFFF, "mean", h[33] // max(h[34], 1), "count", h[34])
eng.close()
