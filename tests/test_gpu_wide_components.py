"""Components of more than 16 haplotypes (missing #3).  The reference has no cap on the haplotypes of a component
(cbdg/graph.cpp:846-924 returns however many walks MaxFlow yields; caller/msa_builder.cpp:29-42 loops over all of them); the
engine's POA kept 16-bit per-edge label masks and refused max_haps > 16.  Round 6: windows whose components all fit 16 haplotypes
take the common kernels, a window with a wider component (17 .. 32, the caller's max_haps) takes the LAB32 kernels in a pass of
its own (poa.hip: launch_msa).  Checked against the oracle, whose POA has no such width anywhere."""
import numpy as np
import pytest

from harness import OracleEngine, compare_asm, compare_geno_calls, compare_vars
from lancet2_amd import capi, synth
from pin_cases import rand_dna

pytestmark = pytest.mark.gpu


def _hand_asm(params, comps_per_window, rng):
    """assembly buffers written by hand: window w holds the components comps_per_window[w] = [n_haplotypes, ...]; every
    component is a random ~950-base REF haplotype + ALT haplotypes that each carry their own subset of a pool of planted
    variants (SNVs, a 3-base deletion, a 5-base insertion, a 24-base deletion: one ALT allele per site)"""
    n = len(comps_per_window)
    asm = capi.alloc_host(capi.asm_out_spec(params, n))
    MC, MH, ML = params.max_comps, params.max_haps, params.max_hap_len
    for w, comps in enumerate(comps_per_window):
        asm["win_ncomp"][w] = len(comps)
        asm["win_k"][w] = 25
        h0 = 0
        for c, nh in enumerate(comps):
            ci = w * MC + c
            ref = rand_dna(rng, int(rng.integers(900, 1001)))
            sites = list(range(40, len(ref) - 60, 55))
            kinds = ["snv", "del3", "ins5", "snv", "del24", "snv"]
            asm["comp_hap0"][ci] = h0
            asm["comp_nhaps"][ci] = nh
            asm["comp_anchor"][ci] = 17 + c
            seen = {ref}
            haps = [ref]
            while len(haps) < nh:
                pick = sorted(rng.choice(len(sites), size=int(rng.integers(1, 5)), replace=False).tolist(), reverse=True)
                b = bytearray(ref)
                for si in pick:
                    p, kind = sites[si], kinds[si % len(kinds)]
                    if kind == "snv":
                        b[p] = b"ACGT"[(b"ACGT".index(bytes([b[p]])) + 1 + si % 3) % 4]
                    elif kind == "del3":
                        del b[p:p + 3]
                    elif kind == "del24":
                        del b[p:p + 24]
                    else:
                        b[p:p] = b"GATTC"
                hb = bytes(b)
                if hb not in seen:
                    seen.add(hb)
                    haps.append(hb)
            for h, seq in enumerate(haps):
                hi = w * MH + h0 + h
                asm["hap_len"][hi] = len(seq)
                asm["hap_bases"][hi * ML:hi * ML + len(seq)] = np.frombuffer(seq, np.uint8)
            h0 += nh
    return asm


@pytest.mark.parametrize("force_lab32", [False, True])
def test_components_of_17_to_32_haplotypes_against_the_oracle(force_lab32, monkeypatch):
    """a batch that mixes ordinary windows with windows of 17 / 20 / 32 haplotypes in one component (and one window with a wide and
    a narrow component): the two passes together -- or, forced, the LAB32 kernels for every window -- give the oracle's variants"""
    from lancet2_amd.engine import Engine
    if force_lab32:
        monkeypatch.setenv("MA_POA_FORCE_LAB32", "1")
    params = capi.default_params(min_k=25, max_k=25, max_haps=32, max_vars=128, max_allele_bytes=4096)
    rng = np.random.default_rng(1720)
    comps = [[3], [20], [2], [32], [17], [16], [24, 5], [4, 3]]
    asm = _hand_asm(params, comps, rng)
    n = len(comps)
    arrs, n2, nr = synth.make_config_batch("C1", n, first_index=77)  # (the POA stage reads only the batch's window count)
    assert n2 == n
    want = OracleEngine(params).msa(arrs, n, nr, asm)
    eng = Engine(params)
    try:
        got = eng.msa(arrs, n, nr, asm)
    finally:
        eng.close()
    bad = compare_vars(params, got, want, n)
    assert not bad, "\n".join(bad[:20])
    assert (want["win_nvars"] >= 1).all() and int(want["win_nvars"].max()) >= 16
    assert int(np.asarray(got["var_hap_allele"]).reshape(n, params.max_vars, 32)[3].any(axis=0).sum()) >= 25  # slots beyond 16 carry alleles


def test_deep_panel_windows_with_more_than_16_haplotypes_end_to_end():
    """the two deep-panel windows of the bench's C4 leg that round 5 left flagged for good (10 056 and 10 098: more than 16
    haplotypes in one component, both also at the traversal cap), re-submitted the way examples/host_driver.cpp does -- a context
    with max_haps = 32 -- through the whole chain: no capacity flag left, every stage equal to the oracle's"""
    from lancet2_amd.engine import Engine
    import bench
    params = capi.default_params(min_k=25, max_k=25, max_haps=32, max_vars=128, max_allele_bytes=8192)
    arrs, n, nr = bench.make_windows("C4", 2, 10_000, 0, 2, indices=[10_056, 10_098])
    orc = OracleEngine(params)
    wa = orc.assemble(arrs, n, nr)
    wv = orc.msa(arrs, n, nr, wa)
    wq = orc.genotype(arrs, n, nr, wa, wv, debug=False)
    MC = params.max_comps
    assert int(wa["comp_nhaps"].reshape(n, MC).max()) > 16, wa["comp_nhaps"].reshape(n, MC).tolist()
    eng = Engine(params)
    try:
        g, a, v, q = eng.process(arrs, n, nr, debug=False)
    finally:
        eng.close()
    over = capi.MA_W_HAP_OVERFLOW | capi.MA_W_LEN_OVERFLOW | capi.MA_W_VAR_OVERFLOW | capi.MA_W_TABLE_OVERFLOW
    assert not (a["win_status"] & np.uint32(over)).any(), a["win_status"].tolist()
    bad = compare_asm(params, a, wa, n) + compare_vars(params, v, wv, n) + compare_geno_calls(q, wq)
    assert not bad, "\n".join(bad[:20])
