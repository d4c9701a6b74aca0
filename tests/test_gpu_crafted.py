"""GPU parity on crafted windows that reach the assembler's rare control-flow branches (test holes): the
complexity gate (graph_complexity.h:112-121), MaxFlow's real 2^20-visit cap (max_flow.h:69), windows of the CLI's
maximum size (-w 2500, core/window_builder.h:25-26) with haplotype slots to match.  The oracle counts the events, so
each test proves that its windows really went where it says."""
import ctypes as C

import numpy as np
import pytest

from harness import OracleEngine, compare_asm, compare_geno, compare_vars, oracle
from lancet2_amd import capi, synth
from pin_cases import many_bubble_window

pytestmark = pytest.mark.gpu


def oracle_events():
    """[cycle found, complexity gate fired, traversal limit hit] since the last call"""
    buf = (C.c_ulonglong * 4)()
    oracle().orc_debug_counters(buf)
    return list(buf)[:3]


@pytest.mark.parametrize("min_k,max_k", [(25, 25), (25, 49)])
def test_complexity_gate_and_traversal_cap(min_k, max_k):
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=min_k, max_k=max_k, max_hap_len=4096)
    wins = [many_bubble_window(9 + i, ns) for i, ns in enumerate((10, 30, 45, 52, 60))]
    arrs, n, nr = synth.pack_batch(wins)
    orc = OracleEngine(params)
    oracle_events()
    wa = orc.assemble(arrs, n, nr)
    cyc, gate, limit = oracle_events()
    assert gate >= 2 and limit >= 2, (cyc, gate, limit)   # 52 / 60 bubbles: gate; 30 / 45: the 2^20 cap
    wv = orc.msa(arrs, n, nr, wa)
    wq = orc.genotype(arrs, n, nr, wa, wv)
    eng = Engine(params)
    try:
        _, a, v, q = eng.process(arrs, n, nr, debug=True)
    finally:
        eng.close()
    bad = compare_asm(params, a, wa, n) + compare_vars(params, v, wv, n)
    bad += compare_geno(params, q, wq, n, nr, wv["win_nvars"], arrs["read_win_off"])
    assert not bad, "\n".join(bad[:20])
    assert (wa["win_status"] & capi.MA_W_LEN_OVERFLOW).sum() == 0
    if max_k > min_k:  # the gated windows come back at a larger k
        assert (wa["win_k"][3:] > 25).all()


def test_maximum_window_size_end_to_end():
    """-w 2500: 2501-base windows with ordinary variant density through gate, assembly, POA and genotyping"""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25, max_hap_len=3072)
    arrs, n, nr = synth.make_config_batch("C2", 2, first_index=424_200, W=2501)
    orc = OracleEngine(params)
    wg = orc.gate(arrs, n, nr)
    wa = orc.assemble(arrs, n, nr)
    wv = orc.msa(arrs, n, nr, wa)
    wq = orc.genotype(arrs, n, nr, wa, wv)
    eng = Engine(params)
    try:
        g, a, v, q = eng.process(arrs, n, nr, debug=True)
    finally:
        eng.close()
    assert np.array_equal(g["max_approx"], wg["max_approx"]) and np.array_equal(g["max_exact"], wg["max_exact"])
    bad = compare_asm(params, a, wa, n) + compare_vars(params, v, wv, n)
    bad += compare_geno(params, q, wq, n, nr, wv["win_nvars"], arrs["read_win_off"])
    assert not bad, "\n".join(bad[:20])
    assert (wa["win_status"] == 0).all() and int(wv["win_nvars"].sum()) >= 4
    assert int(wa["hap_len"].max()) > 2048


@pytest.mark.parametrize("env,cfg,kw", [({"MA_ARENA_CAP": "48"}, "C2", dict(snv_rate=1e-2, indel_rate=2e-3)),
                                         ({"MA_NODE_CAP": "512", "MA_ARENA_CAP": "256"}, "C4", dict(depths=(200, 200))),
                                         # the k-mer table: a first pass planned too small (what deep windows get by design: a
                                         # quarter of their instances), the retry passes plan the full table
                                         ({"MA_TC_FIRST": "11"}, "C2", dict(snv_rate=1e-2, indel_rate=2e-3))])
def test_capacity_overflow_is_retried_inside_the_library(env, cfg, kw, monkeypatch):
    """The reference has one capacity (2^20 BFS visits); the engine's node array and search arena are sized per batch.
    With the capacities forced far too low, a single pass leaves windows flagged TABLE_OVERFLOW (shown with
    MA_NO_CAP_RETRY); the library's own retry passes (4x, 16x) must bring every one of them to the oracle's answer
    with no flag left -- no CPU fallback, no action by the caller."""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25)
    n_win = 12 if cfg == "C2" else 2
    arrs, n, nr = synth.make_config_batch(cfg, n_win, first_index=77_000, **kw)
    orc = OracleEngine(params)
    wa = orc.assemble(arrs, n, nr)
    wv = orc.msa(arrs, n, nr, wa)
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    monkeypatch.setenv("MA_NO_CAP_RETRY", "1")
    eng = Engine(params)
    try:
        a1 = eng.assemble(arrs, n, nr)
    finally:
        eng.close()
    flagged = (a1["win_status"] & capi.MA_W_TABLE_OVERFLOW) != 0
    assert flagged.any(), "the forced capacities do not overflow: the test does not test anything"
    monkeypatch.delenv("MA_NO_CAP_RETRY")
    eng = Engine(params)
    try:
        a = eng.assemble(arrs, n, nr)
        v = eng.msa(arrs, n, nr, a)
    finally:
        eng.close()
    assert not (a["win_status"] & capi.MA_W_TABLE_OVERFLOW).any()
    bad = compare_asm(params, a, wa, n) + compare_vars(params, v, wv, n)
    assert not bad, "\n".join(bad[:10])


def test_capacity_retries_inside_the_ladders_speculative_tail(monkeypatch):
    """Capacity overflow and the k ladder together: with the node array and the search arena forced small, windows overflow
    on the first rungs and in the nested pass that attempts the ladder's tail six rungs at a time (assemble.hip:
    speculate_tail) -- which retries with larger capacities itself, like the outer passes do.  Every window ends on the
    oracle's k with the oracle's haplotypes and no flag left."""
    from lancet2_amd.engine import Engine
    params = capi.default_params()  # the reference's ladder, k = 13 ... 127
    wins = []
    for i, dup in enumerate((40, 70, 0, 100, 0, 55)):
        kw = dict(synth.CONFIGS["C2"])
        if dup:
            kw["tandem_dup"] = dup
        wins.append(synth.make_window(85_000 + i, **kw))
    arrs, n, nr = synth.pack_batch(wins)
    orc = OracleEngine(params)
    wa = orc.assemble(arrs, n, nr)
    monkeypatch.setenv("MA_NODE_CAP", "700")
    monkeypatch.setenv("MA_ARENA_CAP", "64")
    monkeypatch.setenv("MA_NO_CAP_RETRY", "1")
    eng = Engine(params)
    try:
        a1 = eng.assemble(arrs, n, nr)
    finally:
        eng.close()
    assert ((a1["win_status"] & capi.MA_W_TABLE_OVERFLOW) != 0).any(), "the forced capacities do not overflow"
    monkeypatch.delenv("MA_NO_CAP_RETRY")
    eng = Engine(params)
    try:
        a = eng.assemble(arrs, n, nr)
    finally:
        eng.close()
    assert not (a["win_status"] & capi.MA_W_TABLE_OVERFLOW).any()
    bad = compare_asm(params, a, wa, n)
    assert not bad, "\n".join(bad[:10])
    assert (wa["win_k"] > 50).any()


def test_reads_beyond_the_aligner_limit_flag_their_window():
    """reads longer than 608 bases: never a silently truncated alignment, and (since round 3) not a failed batch either --
    the window is assembled, not genotyped, and says so (MA_W_READ_OVERFLOW; the whole story in
    test_over_long_reads_flag_their_window_not_the_batch)"""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25)
    arrs, n, nr = synth.make_config_batch("C2", 1, first_index=424_300, read_len=700, depths=(12, 12))
    eng = Engine(params)
    try:
        _, a, v, q = eng.process(arrs, n, nr)
    finally:
        eng.close()
    assert a["win_status"][0] & capi.MA_W_READ_OVERFLOW
    assert q["allele_counts"].sum() == 0


def test_windows_that_run_out_of_ladder_and_windows_at_different_rungs_share_a_pass():
    """Every window climbs its own k ladder inside one pass: a window whose repeat outlasts the ladder (gated at every rung)
    ends unresolved on the last rung like the reference's loop, next to windows that assemble at 13, 19, ... and to
    tandem-repeat windows that need a larger k."""
    from lancet2_amd.engine import Engine
    params = capi.default_params()  # the reference's ladder 13, 19, ... 127
    wins = [synth.make_window(424_400 + i, **dict(synth.CONFIGS["C2"])) for i in range(3)]
    wins += [synth.make_window(424_410 + i, **dict(synth.CONFIGS["C2"], str_unit=u)) for i, u in enumerate((b"CA", b"GATA", b"AGGGTT"))]
    rep = synth.make_window(424_420, **dict(synth.CONFIGS["C2"]))
    rep["ref"][700:860] = rep["ref"][100:260]  # a 160-base exact repeat: HasExactOrApproxRepeat at every k <= 127
    wins.append(rep)
    arrs, n, nr = synth.pack_batch(wins)
    orc = OracleEngine(params)
    wg, wa = orc.gate(arrs, n, nr), orc.assemble(arrs, n, nr)
    eng = Engine(params)
    try:
        g = eng.gate(arrs, n, nr)
        a = eng.assemble(arrs, n, nr)
    finally:
        eng.close()
    assert np.array_equal(g["max_approx"], wg["max_approx"])
    bad = compare_asm(params, a, wa, n)
    assert not bad, "\n".join(bad[:20])
    assert int(wg["max_approx"][-1]) >= 160 and int(wa["win_k"][-1]) == 127 and (wa["win_status"][-1] & capi.MA_W_NO_HAPLOTYPE)
    assert len(set(int(k) for k in wa["win_k"])) >= 3, wa["win_k"]
