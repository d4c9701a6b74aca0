"""Developer tool: runs tools/dbg/compress_model.cpp (the CPU model of k_clean_chains' first compaction) against the oracle's
sequential CompressGraph on bench-shaped and config-shaped windows.  Build line in the .cpp header."""
import ctypes as C
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import bench  # noqa: E402
from lancet2_amd import capi, synth  # noqa: E402

lib = C.CDLL(os.environ.get("CMODEL_LIB", "/tmp/libcmodel.so"))
names = ["windows", "comps", "mismatched", "punts", "nodes", "alive_after", "segments", "turns", "max_turns", "max_alive",
         "max_deg", "nested", "max_seg", "max_nodes", "noop_turns", "deg_gt4"]


def run(tag, arrs, n, nr, params, k, verbose=1):
    b = capi.make_batch_struct(arrs, n, nr)
    st = (C.c_ulonglong * 16)()
    lib.model_check(C.byref(params), C.byref(b), k, st, verbose)
    d = dict(zip(names, list(st)))
    print(tag, "k=%d" % k, d, flush=True)
    return d


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "bench"
    nwin = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    if what == "bench":
        arrs, n, nr = bench.make_windows("C3", nwin, 10_000, 8, 8)
        p = capi.default_params(min_k=25, max_k=25)
        for k in (25, 31, 43, 61):
            run("bench C3", arrs, n, nr, p, k)
    else:
        for cfg in ("C1", "C2", "C4", "C5"):
            ns = 3 if cfg == "C5" else 2
            arrs, n, nr = synth.make_config_batch(cfg, nwin if cfg != "C4" else max(2, nwin // 16))
            p = capi.default_params(min_k=25, max_k=25, num_samples=ns) if ns != 2 else capi.default_params(min_k=25, max_k=25)
            for k in (13, 25, 37):
                run(cfg, arrs, n, nr, p, k)
