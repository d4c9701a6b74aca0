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
