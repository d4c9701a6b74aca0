"""Pins the ORACLE against every known-answer vector the reference's own tests hold for the hot
path.  Values are transcribed from the cited reference tests."""
import ctypes as C

import numpy as np
import pytest

from harness import oracle

BASE_SEED = 0x5EED5EED5EED5EED


def b(s):
    return s.encode() if isinstance(s, str) else s


def hamming(a, c):
    return oracle().orc_hamming(b(a), b(c), C.c_uint64(len(a)))


def has_repeat(kmers_as_seq, k, mm):
    return bool(oracle().orc_has_repeat(b(kmers_as_seq), C.c_uint64(len(kmers_as_seq)), C.c_uint64(k), C.c_uint64(mm)))


def revcomp(s):
    out = C.create_string_buffer(len(s))
    oracle().orc_revcomp(b(s), C.c_uint64(len(s)), out)
    return out.raw.decode()


# ---- tests/base/repeat_test.cpp:76-157 -------------------------------------------------------
def test_hamming_small():
    assert hamming("aaaa", "aaaa") == 0
    assert hamming("aaaa", "abaa") == 1
    assert hamming("aaaa", "aaba") == 1
    assert hamming("abaa", "aaba") == 2


def test_hamming_simd_boundaries():
    assert hamming("A" * 32, "A" * 32) == 0
    assert hamming("A" * 32, "C" * 32) == 32
    lhs, rhs = "A" * 33, "A" * 32 + "T"
    assert hamming(lhs, rhs) == 1
    assert hamming("C" + lhs[1:], "A" + rhs[1:]) == 2
    r31 = list("A" * 31)
    r31[10] = "T"
    assert hamming("A" * 31, "".join(r31)) == 1
    assert hamming("A", "A") == 0 and hamming("A", "T") == 1
    assert hamming("", "") == 0


def _has_repeat_kmers(kmers, mm):
    """The reference KATs pass explicit k-mer lists; emulate with a sequence whose sliding k-mers
    are exactly those k-mers is not possible in general, so check pairwise with the oracle's
    Hamming distance (HasRepeat == any pair within mm: base/repeat.cpp:348-371)."""
    n = len(kmers)
    if mm == 0:
        return len(set(kmers)) != n
    return any(hamming(kmers[i], kmers[j]) <= mm for i in range(n) for j in range(i + 1, n))


def test_has_repeat_kats():  # tests/base/repeat_test.cpp:163-212
    assert _has_repeat_kmers(["ACGT", "TGCA", "ACGT", "GGCC"], 0)
    assert not _has_repeat_kmers(["ACGT", "TGCA", "GGCC", "AATT"], 0)
    assert not _has_repeat_kmers([], 0) and not _has_repeat_kmers(["ACGT"], 0)
    assert _has_repeat_kmers(["ACGT", "TGCA", "ACGA"], 1)
    assert not _has_repeat_kmers(["ACGT", "TGCA", "ACGA"], 0)
    assert not _has_repeat_kmers(["AAAA", "CCCC", "GGGG", "TTTT"], 1)


def test_has_repeat_on_sequences():
    # sliding-window form used by the graph gate (cbdg/graph.h:127-131)
    seq = "ACGTTGCAACGTAGGC"  # 4-mer ACGT occurs at 0 and 8
    assert has_repeat(seq, 4, 0)
    assert not has_repeat("ACGTTGCA", 4, 0)
    assert not has_repeat("ACG", 4, 0)  # fewer than one k-mer
    # approximate: ACGTA.. vs ACGAA..
    assert has_repeat("ACGTACCCCACGAAC", 5, 1)
    assert not has_repeat("ACGTACCCCACGAAC", 5, 0)


def test_has_repeat_monotone_in_k():
    rng = np.random.default_rng(7)
    for it in range(5):
        s = "".join("ACGT"[i] for i in rng.integers(0, 4, 300))
        s = s[:100] + s[20:60] + s[100:]  # plant an exact 40 bp repeat
        prev = True
        for k in range(2, 80):
            cur = has_repeat(s, k, 2)
            assert prev or not cur
            prev = cur
        assert has_repeat(s, 40, 0) and has_repeat(s, 42, 2)


# ---- tests/base/rev_comp_test.cpp:15-119 (behaviour of the table) ------------------------------
def test_revcomp_table():
    assert revcomp("ACGT") == "ACGT"
    assert revcomp("AACC") == "GGTT"
    assert revcomp("acgtN") == "Nacgt"
    assert revcomp("AXG") == "CNT"  # non-ACGT -> N
    assert revcomp("") == ""


# ---- hts/phred_quality.cpp:15 spot-check anchors ------------------------------------------------
def test_phred_anchors():
    o = oracle()
    assert o.orc_phred(0) == 1.0 and o.orc_phred(10) == 0.1 and o.orc_phred(20) == 0.01
    assert o.orc_phred(30) == 0.001 and o.orc_phred(40) == 0.0001
    assert o.orc_phred(1) == 0.7943282347242815


# ---- tests/base/compute_stats_test.cpp (Median / Welford semantics) ---------------------------
def test_median_and_stats():
    o = oracle()

    def med(v):
        a = np.array(v, dtype=np.uint32)
        return o.orc_median_u32(a.ctypes.data_as(C.c_void_p), C.c_uint64(len(a)))

    assert med([]) == 0 and med([7]) == 7
    assert med([1, 2, 3]) == 2 and med([3, 1, 2]) == 2
    assert med([1, 2, 3, 4]) == 2  # integer (2+3)/2
    assert med([10, 20]) == 15
    v = np.array([2.0, 4.0, 4.0, 4.0, 5.0, 5.0, 7.0, 9.0])
    m, var, sd = C.c_double(), C.c_double(), C.c_double()
    o.orc_online_stats(v.ctypes.data_as(C.c_void_p), C.c_uint64(len(v)), C.byref(m), C.byref(var), C.byref(sd))
    assert m.value == 5.0 and abs(var.value - 32.0 / 7.0) < 1e-12 and abs(sd.value - np.sqrt(32.0 / 7.0)) < 1e-12


# ---- tests/hts/cigar_utils_test.cpp:58-172 -------------------------------------------------------
def _cig(txt):
    import re
    ops = re.findall(r"(\d+)([MIDNSHP=X])", txt)
    return "".join(o for _, o in ops).encode(), np.array([int(n) for n, _ in ops], dtype=np.uint32)


def edit_distance(cigar, q, t):
    ops, lens = _cig(cigar)
    qa, ta = np.array(q, dtype=np.uint8), np.array(t, dtype=np.uint8)
    return oracle().orc_edit_distance(ops, lens.ctypes.data_as(C.c_void_p), len(lens),
                                      qa.ctypes.data_as(C.c_void_p), len(qa),
                                      ta.ctypes.data_as(C.c_void_p), len(ta))


def refpos_to_qpos(cigar, ref_pos):
    ops, lens = _cig(cigar)
    return oracle().orc_refpos_to_qpos(ops, lens.ctypes.data_as(C.c_void_p), len(lens), C.c_uint64(ref_pos))


def test_edit_distance_kats():
    assert edit_distance("4M", [0, 1, 2, 3], [0, 1, 2, 3]) == 0
    assert edit_distance("4M", [0, 1, 2, 3], [0, 1, 0, 3]) == 1
    assert edit_distance("2M2I2M", [0, 1, 3, 3, 2, 3], [0, 1, 2, 3]) == 2
    assert edit_distance("2M2D2M", [0, 1, 2, 3], [0, 1, 3, 3, 2, 3]) == 2
    assert edit_distance("2S4M", [3, 3, 0, 1, 2, 3], [0, 1, 2, 3]) == 0  # clips excluded from NM
    assert edit_distance("2=1X1=", [0, 1, 0, 3], [0, 1, 2, 3]) == 1
    assert edit_distance("1M1I1M1D1M", [0, 3, 1, 2], [0, 1, 3, 2]) == 2


def test_refpos_to_qpos_kats():
    assert refpos_to_qpos("10M", 0) == 0 and refpos_to_qpos("10M", 5) == 5
    assert refpos_to_qpos("3M2I3M", 3) == 5   # insertion shifts the query
    assert refpos_to_qpos("3M2D3M", 3) == 3   # inside the deletion -> query pos at its start
    assert refpos_to_qpos("3M2D3M", 4) == 3
    assert refpos_to_qpos("3M2D3M", 5) == 3
    assert refpos_to_qpos("2S4M", 0) == 2     # soft clip advances the query only
    assert refpos_to_qpos("4M", 10) == 4      # beyond the CIGAR -> end of query


# ---- tests/caller/variant_set_test.cpp:35-247: the only reference tests that run an aligner -------
def poa_variants(seqs, anchor=100, eng=(3, -5, -3, -3, -3, -3)):
    buf = b"\0".join(s.encode() for s in seqs) + b"\0"
    out = C.create_string_buffer(1 << 16)
    n = oracle().orc_poa_variants(buf, len(seqs), *[C.c_int(x) for x in eng], C.c_uint64(anchor), out, len(out))
    assert n >= 0
    vs = []
    for line in out.value.decode().splitlines():
        pos, ref, alts, haps, rs = line.split("\t")
        vs.append(dict(pos=int(pos), ref=ref, alts=alts.split(","),
                       haps=[[tuple(map(int, x.split(":"))) for x in h.split(",")] for h in haps.split(";")],
                       ref_start=int(rs)))
    return vs


def test_poa_kat_snv():
    vs = poa_variants(["ATCG", "AGCG"])
    assert len(vs) == 1 and vs[0]["ref"] == "T" and vs[0]["alts"] == ["G"] and vs[0]["pos"] == 101


def test_poa_kat_deletion():
    vs = poa_variants(["ATCG", "AG"])
    assert len(vs) == 1 and vs[0]["ref"] == "ATC" and vs[0]["alts"] == ["A"] and vs[0]["pos"] == 100


def test_poa_kat_overlapping_multiallelic():
    vs = poa_variants(["ATGTGC", "ACGTGC", "AGC", "ATGTAC"])
    assert len(vs) == 1 and len(vs[0]["alts"]) == 3
    carried = {h for hl in vs[0]["haps"] for (h, _s) in hl}
    assert carried == {1, 2, 3}


def test_poa_kat_insertion():
    vs = poa_variants(["ATCG", "ATAACG"])
    assert len(vs) == 1 and vs[0]["ref"] == "T" and vs[0]["alts"] == ["TAA"] and vs[0]["pos"] == 101


def test_poa_kat_mnp():
    vs = poa_variants(["ATCG", "AAAG"])
    assert len(vs) == 1 and vs[0]["ref"] == "TC" and vs[0]["alts"] == ["AA"]


def test_poa_kat_complex_split():
    vs = poa_variants(["ATCG", "AAAAG"])
    assert len(vs) == 2
    assert vs[0]["pos"] == 100 and vs[0]["ref"] == "" and vs[0]["alts"][0] == "A"
    assert vs[1]["pos"] == 101 and vs[1]["ref"] == "TC" and vs[1]["alts"] == ["AA"]


def test_poa_kat_nway_sink():
    vs = poa_variants(["ATCG", "AGCG", "AACG", "ACG"])
    assert len(vs) == 1 and vs[0]["ref"] == "AT" and sorted(vs[0]["alts"]) == ["A", "AA", "AG"]


# ---- tests/cbdg/kmer_test.cpp:98-248 property: merging adjacent k-mers reproduces the sequence ----
def _merge_chain(seq, k, reverse):
    out = C.create_string_buffer(len(seq) + 8)
    n = oracle().orc_kmer_merge_chain(seq.encode(), C.c_uint64(len(seq)), C.c_uint64(k), reverse, out, C.c_uint64(len(out)))
    assert n >= 0
    return out.value.decode()


@pytest.mark.parametrize("k,length,iters", [(11, 12, 200), (21, 1024, 20), (25, 151, 50), (127, 600, 5)])
def test_kmer_merge_property(k, length, iters):
    rng = np.random.default_rng(BASE_SEED % (1 << 32))
    for _ in range(iters):
        s = "".join("ACGT"[i] for i in rng.integers(0, 4, length))
        for rev in (0, 1):
            m = _merge_chain(s, k, rev)
            assert m in (s, revcomp(s))


def test_confidence_examples():  # src/lancet/cbdg/node.cpp:53-58
    def conf(counts, ns, ref):
        a = np.array(counts, dtype=np.uint32)
        return oracle().orc_confidence(a.ctypes.data_as(C.c_void_p), len(a), ns, int(ref))

    assert conf([20, 18], 2, True) == 39
    assert conf([0, 15], 2, False) == 7
    assert conf([1, 1], 2, True) == 1      # singleton override
    assert conf([1, 0], 2, False) == 1
    assert conf([0, 0], 2, True) == 0
    assert conf([3], 2, False) == 1        # lazily-sized counts: floor(3 * 1/2)


# ==== SURVEY 8 f3: sequence-complexity annotation ==============================================
def max_hrun(s):
    return oracle().orc_max_hrun(b(s), C.c_uint64(len(s)))


def entropy(s):
    return float(oracle().orc_entropy(b(s), C.c_uint64(len(s))))


def find_repeats(s, approx):
    out = np.zeros(5 * 256, np.int32)
    cp = np.zeros(256, np.float32)
    n = oracle().orc_find_repeats(b(s), C.c_uint64(len(s)), int(approx), out.ctypes.data_as(C.c_void_p),
                                  cp.ctypes.data_as(C.c_void_p), 256)
    return [dict(period=int(out[5 * i]), start=int(out[5 * i + 1]), span=int(out[5 * i + 2]),
                 errors=int(out[5 * i + 3]), exact=bool(out[5 * i + 4]), copies=float(cp[i])) for i in range(n)]


def longdust(s, k=7, max_len=1024, gc=0.41, one_strand=False):
    return float(oracle().orc_longdust(b(s), C.c_uint64(len(s)), k, max_len, C.c_double(gc), int(one_strand)))


def seqcx(ref, rpos, rlen, alt, apos, alen, gc=0.41):
    oi, of, od = np.zeros(4, np.int32), np.zeros(4, np.float32), np.zeros(3, np.float64)
    oracle().orc_seqcx_score(b(ref), C.c_uint64(len(ref)), C.c_uint64(rpos), C.c_uint64(rlen), b(alt),
                             C.c_uint64(len(alt)), C.c_uint64(apos), C.c_uint64(alen), C.c_double(gc),
                             oi.ctypes.data_as(C.c_void_p), of.ctypes.data_as(C.c_void_p),
                             od.ctypes.data_as(C.c_void_p))
    return dict(ctx_hrun=int(oi[0]), delta_hrun=int(oi[1]), tr_period=int(oi[2]), stutter=int(oi[3]),
                ctx_entropy=float(of[0]), delta_entropy=float(of[1]), tr_affinity=float(of[2]),
                tr_purity=float(of[3]), ctx_flank_lq=float(od[0]), ctx_hap_lq=float(od[1]),
                delta_flank_lq=float(od[2]))


# ---- tests/base/sequence_complexity_test.cpp:16-28 ----
def test_max_homopolymer_run():
    assert max_hrun("") == 0
    assert max_hrun("A") == 1
    assert max_hrun("ACGT") == 1
    assert max_hrun("AACCCGTTT") == 3
    assert max_hrun("AAAAAAA") == 7
    assert max_hrun("ATCAAAAAGTC") == 5
    assert max_hrun("T" * 50) == 50


# ---- tests/base/sequence_complexity_test.cpp:35-53 ----
def test_local_shannon_entropy():
    assert entropy("") == 0.0
    assert entropy("AAAA") == 0.0
    assert entropy("TTTTTTTT") == 0.0
    assert entropy("ACGT") == pytest.approx(2.0, abs=1e-3)
    assert entropy("AACCGGTT") == pytest.approx(2.0, abs=1e-3)
    assert entropy("ACACAC") == pytest.approx(1.0, abs=1e-3)
    assert entropy("AACCGG") == pytest.approx(np.log2(3.0), abs=1e-2)


# ---- tests/base/sequence_complexity_test.cpp:60-117 ----
def test_find_exact_repeats():
    rs = find_repeats("ATATATATAT", approx=False)
    assert rs
    best = rs[0]
    for r in rs:
        if r["copies"] > best["copies"] or (r["copies"] == best["copies"] and r["period"] < best["period"]):
            best = r
    assert best["period"] == 2 and best["copies"] == pytest.approx(5.0, abs=0.01)
    assert best["span"] == 10 and best["exact"] and best["errors"] == 0
    rs = find_repeats("AAAAAA", approx=False)
    hit = [r for r in rs if r["period"] == 1 and r["copies"] >= 6.0]
    assert hit and all(r["span"] == 6 and r["exact"] for r in hit)
    assert all(r["period"] > 0 for r in find_repeats("ACGTACGA", approx=False))
    # primitive-motif enforcement: ATAT must never be reported with period 4
    assert all(r["period"] != 4 for r in find_repeats("ATATATAT", approx=False))


# ---- tests/base/sequence_complexity_test.cpp:120-145 ----
def test_find_approx_repeats():
    rs = find_repeats("CAGCAACAGCAG", approx=True)
    found = False
    for r in rs:
        assert r["period"] >= 1 and r["copies"] >= 1.0 and r["span"] >= r["period"] and r["errors"] >= 0
        purity = 1.0 - r["errors"] / r["span"]
        assert 0.0 <= purity <= 1.0
        if r["period"] == 3 and r["copies"] >= 3.0:
            found = True
            assert r["errors"] >= 1 and purity >= 0.75
    assert found


# ---- tests/base/sequence_complexity_test.cpp:152-222 ----
def test_seqcx_score_cases():
    ref = "C" * 90 + "A" * 20 + "G" * 90
    alt = "C" * 90 + "A" * 25 + "G" * 85
    c = seqcx(ref, 90, 20, alt, 90, 25)
    assert c["ctx_hrun"] >= 20 and c["ctx_entropy"] >= 0.0 and c["delta_hrun"] >= 0
    assert c["ctx_flank_lq"] >= 0.0 and c["ctx_hap_lq"] >= 0.0
    hap = "ACGT" * 50
    c = seqcx(hap, 100, 1, hap, 100, 1)
    assert c["ctx_hrun"] == 1 and c["ctx_entropy"] == pytest.approx(2.0, abs=0.1)
    assert c["delta_hrun"] == 0 and c["delta_entropy"] == pytest.approx(0.0, abs=0.01)
    hap = "ACGTACGTACGTACGTACGTACGTACGTACGT" + "TGCATGCATGCATGCATGCATGCATGCATGCA"
    c = seqcx(hap, 16, 1, hap, 16, 1)
    assert 0.0 <= c["tr_affinity"] <= 1.0 and c["tr_purity"] >= 0.0 and c["tr_period"] >= 0 and c["stutter"] >= 0
    hap = "A" * 200
    c = seqcx(hap, 100, 1, hap, 100, 1, gc=0.5)
    assert c["ctx_hap_lq"] == pytest.approx(np.log1p(longdust(hap, 7, 4096, 0.5)), abs=1e-4)


# ---- tests/base/sequence_complexity_test.cpp:229-271 (GC bias) ----
def test_longdust_gc_bias():
    rnd = "GCTAAGGTCCTTGAACGGATTCATAGCCTGAGATTTCAAC" "TGCAAGGTCCTCATGAACTTTAGCCCAAGATTCTGAACGT"
    assert longdust(rnd, gc=0.5) < 0.5 and longdust(rnd, gc=0.41) < 0.5
    at_rich = "ATATATGTAACTTAATGTATTATATTGATGAATTTAATGG" "ATTAAGTCATATTAATGATTAATATGATATAAGAAATAGG"
    assert longdust(at_rich, gc=0.41) <= longdust(at_rich, gc=0.5)
    assert longdust("A" * 50) > 0.5 and longdust("CA" * 25) > 0.5
    assert longdust("A" * 100, gc=0.5) > 0.0


# ---- tests/base/longdust_scorer_test.cpp:296-351 ----
def test_longdust_properties():
    for s in ("", "ATCG", "ATCGAT", "N" * 19):
        assert longdust(s) == 0.0
    assert longdust("A" * 10) > 0.0 and longdust("A" * 20) > 0.6 and longdust("A" * 50) > 1.0
    assert longdust("A" * 20) < longdust("A" * 50) < longdust("A" * 100)
    s5, s10, s20 = (longdust("TTAGGG" * c) for c in (5, 10, 20))
    assert s5 < s10 <= s20
    assert longdust("T" * 30) > 0.6 and longdust("T" * 30) >= longdust("T" * 30, one_strand=True)
    assert longdust("TTAGGG" * 20) == pytest.approx(longdust("ttaggg" * 20))
    assert longdust("AAAAAAANAAAAAAA") < longdust("A" * 15)
    rng = np.random.default_rng(42)
    for n in (100, 200, 500):
        assert longdust("".join("ACGT"[i] for i in rng.integers(0, 4, n))) < 0.1


# ---- tests/base/longdust_scorer_test.cpp:385-430, 582-610 ----
def test_longdust_ftable_shape_and_monotone_homopolymers():
    for k, ml, gc in ((7, 1024, 0.41), (7, 256, 0.5), (7, 64, 0.5), (7, 512, 0.5)):
        out = np.zeros(ml + 1, np.float64)
        n = oracle().orc_longdust_ftable(k, ml, C.c_double(gc), out.ctypes.data_as(C.c_void_p))
        assert n == ml + 1 and out[0] == 0.0
        assert np.all(out >= 0.0) and np.all(np.diff(out) >= 0.0)
    for base in "ACGT":
        prev = 0.0
        for copies in range(7, 51):
            sc = longdust(base * copies)
            assert sc >= prev - 1e-9
            prev = sc


def test_longdust_closed_form_for_homopolymers():
    """Independent restatement (python, math.lgamma): poly-A of n bases at k=7, uniform GC has one 7-mer with
    count c = n - 6, so Q = (lgamma(c + 1) - 4^7 f1(c / 4^7)) / c with f1 the Poisson series of
    longdust_scorer.h:350-389."""
    import math

    def f1(lam):
        acc, sum_n, scaled = 0.0, 0.0, lam
        for cnt in range(2, 10001):
            sum_n += math.log(cnt)
            scaled *= lam / cnt
            z = scaled * sum_n
            if z < acc * 1e-9:
                break
            acc += z
        return acc * math.exp(-lam)

    for n in (10, 20, 50, 100, 300):
        c = n - 6
        want = max(0.0, (math.lgamma(c + 1) - 16384 * f1(c / 16384)) / c)
        assert longdust("A" * n, 7, 1024, 0.5) == pytest.approx(want, rel=1e-12)


def test_graph_complexity_chain_and_single_bubble():
    """tests/cbdg/graph_test.cpp:96-205: a linear chain has cyclomatic complexity E - V + 1 = 0, no branch point and
    at most one edge per direction; a single bubble has M = 1, a maximum single-direction degree of 2 and at least
    one branch point.  Here through the whole cleaning path: a window without a variant prunes down to a chain
    (source, middle, sink), a window with one somatic SNV to exactly one bubble."""
    from harness import OracleEngine
    from lancet2_amd import capi, synth
    p = capi.default_params(min_k=25, max_k=25)
    clean = dict(snv_rate=1e-9, indel_rate=0.0, error_scale=0.0)
    for n_somatic, want_m, want_deg in ((0, 0, 1), (1, 1, 2)):
        arrs, n, nr = synth.make_config_batch("C2", 3, first_index=5000, n_somatic=n_somatic, **clean)
        a = OracleEngine(p).assemble(arrs, n, nr)
        cx = a["comp_cx"].reshape(n, p.max_comps, 3)[:, 0]
        assert a["win_ncomp"].tolist() == [1, 1, 1]
        for cyclomatic, branch_points, max_dir in cx.tolist():
            assert cyclomatic == want_m
            assert max_dir <= want_deg if n_somatic == 0 else max_dir == want_deg
            assert (branch_points == 0) if n_somatic == 0 else (branch_points >= 1)
        assert a["comp_nhaps"].reshape(n, p.max_comps)[:, 0].tolist() == [1 + n_somatic] * 3


def test_genotype_pls_against_scipy():
    """caller/genotype_likelihood.cpp:93-272 -- the oracle's Dirichlet-multinomial PLs equal an independent scipy
    evaluation of ln P(c | alpha) = lnG(sum a) - lnG(N + sum a) + sum [lnG(c_i + a_i) - lnG(a_i)] with
    alpha = M mu, M = (1 - 0.01) / 0.01, mu = eps / K background + (1 - eps) on the genotype's alleles, eps = 0.005;
    plus the properties the VCF relies on (best genotype has PL 0, GQ = second-smallest PL capped at 99)."""
    import ctypes as C
    from scipy.special import gammaln
    lib = oracle()
    rng = np.random.default_rng(5)

    def ref_pls(counts):
        K = len(counts)
        M, eps = (1.0 - 0.01) / 0.01, 0.005
        lls = []
        for b in range(K):
            for a in range(b + 1):
                mu = np.full(K, eps / K)
                if a == b:
                    mu[a] += 1.0 - eps
                else:
                    mu[a] += (1.0 - eps) / 2.0
                    mu[b] += (1.0 - eps) / 2.0
                al = np.maximum(1e-6, mu * M)
                c = np.asarray(counts, dtype=np.float64)
                lls.append(float(gammaln(al.sum()) - gammaln(c.sum() + al.sum()) + (gammaln(c + al) - gammaln(al)).sum()))
        lls = np.array(lls)
        return np.round(np.minimum(-10.0 * (lls - lls.max()) / np.log(10.0), 2 ** 31)).astype(np.int64), lls

    cases = [[30, 0], [0, 30], [15, 15], [28, 2], [0, 0], [500, 480], [2000, 3], [10, 10, 10], [40, 0, 7, 1], [3, 0, 0, 0, 9]]
    cases += [list(rng.integers(0, int(rng.choice([5, 40, 400])), int(rng.integers(2, 6)))) for _ in range(200)]
    for counts in cases:
        K = len(counts)
        pls = (C.c_uint32 * 32)()
        gq = C.c_uint32()
        n = lib.orc_genotype_pls((C.c_int32 * K)(*[int(x) for x in counts]), K, pls, C.byref(gq))
        assert n == K * (K + 1) // 2
        got = np.array(pls[:n], dtype=np.int64)
        want, lls = ref_pls(counts)
        # integers: equal unless a likelihood sits on a rounding boundary to within the two libms' difference
        raw = -10.0 * (lls - lls.max()) / np.log(10.0)
        near = np.abs(raw - np.floor(raw) - 0.5) < 1e-6
        assert np.array_equal(got[~near], want[~near]), (counts, got, want)
        assert got.min() == 0
        srt = np.sort(got)
        assert gq.value == min(int(srt[1] - srt[0]), 99)
    # hom-ref evidence calls 0/0, balanced evidence calls 0/1, hom-alt evidence calls 1/1
    for counts, best in (([30, 0], 0), ([15, 15], 1), ([0, 30], 2)):
        pls = (C.c_uint32 * 8)()
        gq = C.c_uint32()
        lib.orc_genotype_pls((C.c_int32 * 2)(*counts), 2, pls, C.byref(gq))
        assert int(np.argmin(pls[:3])) == best
