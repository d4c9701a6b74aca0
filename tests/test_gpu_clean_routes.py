"""The three routes through graph cleaning (round 4): k_clean_chains + k_clean_tail on the compact graph (LDS images of two
sizes, or the arrays in HBM), and k_clean from the raw graph.  Which route a window takes depends on its size and shape
only; the results must not.  Every case is checked against the oracle AND against the other routes."""
import numpy as np
import pytest

from harness import OracleEngine, compare_asm
from lancet2_amd import capi, synth
from lancet2_amd.engine import Engine

pytestmark = pytest.mark.gpu

ASM_KEYS = ("win_status", "win_k", "win_ncomp", "comp_anchor", "comp_hap0", "comp_nhaps", "comp_cx", "comp_cxf", "hap_len",
            "hap_nruns", "hap_stats")


def assemble(params, arrs, n, nr, kernels=None):
    eng = Engine(params)
    try:
        eng.set_streams(1)
        out = eng.assemble(arrs, n, nr)
        if kernels is not None:
            for k, ms in eng.kernel_times():
                kernels[k] = kernels.get(k, 0.0) + ms
        return out
    finally:
        eng.close()


CASES = [
    ("C2", 10, 40_000, {}, dict(min_k=25, max_k=25)),
    ("C3", 6, 41_000, dict(error_scale=4.0), dict(min_k=25, max_k=25)),                       # many error bubbles and tips
    ("C2", 8, 42_000, dict(snv_rate=1e-2, indel_rate=2e-3), dict(min_k=25, max_k=25)),         # dense variants: many segments
    ("C3", 6, 43_000, dict(tandem_dup=40, softclip_frac=0.03), {}),                            # cycles: the k ladder, larger k
    ("C5", 4, 44_000, {}, dict(min_k=25, max_k=25, num_samples=3)),                            # three averaged counts per node
    ("C2", 4, 45_000, dict(W=2501), dict(min_k=25, max_k=25, max_hap_len=4096)),               # beyond the small size class
    ("C2", 6, 46_000, dict(str_unit=b"CA", low_complexity=80), {}),
]


@pytest.mark.parametrize("cfg,nwin,first,kw,pk", CASES)
def test_compact_and_raw_routes_agree(cfg, nwin, first, kw, pk, monkeypatch):
    params = capi.default_params(**pk)
    arrs, n, nr = synth.make_config_batch(cfg, nwin, first_index=first, **kw)
    want = OracleEngine(params).assemble(arrs, n, nr)
    outs = {}
    for route, env in (("compact", {}), ("raw", {"MA_NO_CHAINS": "1"}), ("large classes", {"MA_CHAINS_CAP": "64"})):
        for k in ("MA_NO_CHAINS", "MA_CHAINS_CAP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        outs[route] = assemble(params, arrs, n, nr)
        bad = compare_asm(params, outs[route], want, n)
        assert not bad, route + ": " + "\n".join(bad[:10])
    for k in ASM_KEYS:
        assert np.array_equal(outs["compact"][k], outs["raw"][k]), k


def test_the_compact_route_is_the_one_taken(monkeypatch):
    """the bench's windows never need k_clean: it only finds its early exit (a few microseconds per launch)"""
    monkeypatch.delenv("MA_NO_CHAINS", raising=False)
    params = capi.default_params(min_k=25, max_k=25)
    arrs, n, nr = synth.make_config_batch("C3", 32, first_index=47_000)
    kernels = {}
    assemble(params, arrs, n, nr, kernels)
    assert "k_clean_chains" in kernels and "k_clean_tail" in kernels
    assert kernels["k_clean"] < 0.2 * (kernels["k_clean_chains"] + kernels["k_clean_tail"]), kernels


def test_deep_panel_windows_incl_the_traversal_limit():
    """C4 at full depth (500x/500x, 50 bp indels): graphs of ~3000 nodes, nodes with five and more edges, compact graphs of
    ~1000 nodes, components the reference searches up to its 2^20-pop cap (window 91 003 reports BFS_LIMIT and still yields
    haplotypes: the cap falls inside a level that holds a qualifying arrival)."""
    params = capi.default_params(min_k=25, max_k=25)
    arrs, n, nr = synth.make_config_batch("C4", 4, first_index=91_000)
    want = OracleEngine(params).assemble(arrs, n, nr)
    assert int(want["win_status"][3]) & capi.MA_W_BFS_LIMIT
    got = assemble(params, arrs, n, nr)
    bad = compare_asm(params, got, want, n)
    assert not bad, "\n".join(bad[:10])
    assert not (got["win_status"] & capi.MA_W_TABLE_OVERFLOW).any()
