"""Pins for the two restated third-party aligners ("parity unpinned" for SPOA's convex model and for
minimap2).  The oracle's optimal SCORES are compared with tests/brute_force.py -- numpy, different formulation,
no shared code -- on seeded random cases and on committed golden vectors (tests/golden/pins_aligner.npz, made by
tests/golden/make_aligner_pins.py from the brute force alone).  Tie rules (which of several co-optimal alignments
is reported) stay documented conventions; what is pinned here is optimality and internal consistency."""
import ctypes as C
import os

import numpy as np
import pytest

import brute_force as bf
from harness import oracle
from pin_cases import CASES, make_haplotypes, make_pair, rand_dna

GOLD = os.path.join(os.path.dirname(__file__), "golden", "pins_aligner.npz")


def oracle_align(read, hap, min_score=80):
    lib = oracle()
    rec = (C.c_int32 * 6)()
    txt = C.create_string_buffer(8192)
    rc = lib.orc_align_pair(read, len(read), hap, len(hap), min_score, rec, txt, 8192)
    assert rc == 0
    cig, num = [], ""
    for ch in txt.value.decode():
        if ch.isdigit():
            num += ch
        else:
            cig.append((ch, int(num)))
            num = ""
    return list(rec), cig


def check_pair(read, hap, want_hit=None, want_score=None):
    rec, cig = oracle_align(read, hap)
    hit, score = bf.canonical_pair(read, hap) if want_hit is None else (want_hit, want_score)
    assert bool(rec[0]) == bool(hit), (rec, hit, score)
    if not hit:
        return rec, cig
    assert rec[1] == score, f"oracle score {rec[1]} != brute-force optimum {score}"
    # the reported alignment is worth what it claims and spans what it claims
    s, qp, tp = bf.cigar_score(read, hap, rec[2], cig)
    assert s == rec[1] and qp == len(read) and tp == rec[3], (s, qp, tp, rec, cig)
    lead = cig[0][1] if cig[0][0] == "S" else 0
    trail = cig[-1][1] if len(cig) > 1 and cig[-1][0] == "S" else 0
    assert lead == rec[4] and len(read) - trail == rec[5]
    # soft clips only where the read hangs over a haplotype end (overlap alignment, end bonus in the reference)
    assert lead == 0 or rec[2] == 0
    assert trail == 0 or rec[3] == len(hap)
    return rec, cig


@pytest.mark.parametrize("case", CASES)
def test_read_aligner_scores_are_optimal(case):
    rng = np.random.default_rng(1234 + CASES.index(case))
    for _ in range(60):
        read, hap = make_pair(rng, case)
        check_pair(read, hap)


def test_read_aligner_golden_pins():
    g = np.load(GOLD)
    reads, haps = g["reads"], g["haps"]
    ro, ho = g["read_off"], g["hap_off"]
    for x in range(len(g["hit"])):
        read, hap = bytes(reads[ro[x]:ro[x + 1]]), bytes(haps[ho[x]:ho[x + 1]])
        check_pair(read, hap, bool(g["hit"][x]), int(g["score"][x]))


def test_search_region_never_cuts_a_supported_hit():
    """when the optimum inside R is a hit whose cost leaves more exact 11-mers on the path than
    there are seeds outside the region's core, the UNRESTRICTED optimum is the same alignment score -- the region is
    an optimisation of the search, like the reference's bw = 10000, not a semantic limit."""
    rng = np.random.default_rng(99)
    checked = 0
    for _ in range(120):
        read, hap = make_pair(rng, str(rng.choice(["clean", "noisy", "indel"])))
        hit, score = bf.canonical_pair(read, hap)
        if not hit:
            continue
        m = len(read)
        cost = m - score
        if m - 10 - 11 * (cost // 5) - cost % 5 <= 0:
            continue  # too noisy for the seed argument: no claim
        free, _, _ = bf.overlap_best(read, hap)
        sd = bf.seed_diagonals(read, hap, bf.reach(m))
        if sd[1] - sd[0] > 64:
            continue  # spurious far seeds: the argument needs the vote counts, not exercised here
        assert free == score, (free, score)
        checked += 1
    assert checked > 40


def test_long_indels_follow_the_cost_bound():
    """A 150-base read across a 100-300 bp deletion cannot score 80 end to end (cost 12 + 3 L); one that ends inside
    the flank aligns to the side it mostly lies on; a 250-base read across a 40 bp indel still aligns through it."""
    rng = np.random.default_rng(7)
    for L in (100, 200, 300):
        hap = rand_dna(rng, 1200)
        alt = hap[:500] + hap[500 + L:]
        read = alt[425:575]  # 75 bases either side of the deletion
        rec, _ = check_pair(read, hap)
        assert rec[0] == 0
        rec, cig = check_pair(read, alt)
        assert rec[0] == 1 and rec[1] == 150 and cig == [("M", 150)]
    hap = rand_dna(rng, 1000)
    alt = hap[:400] + hap[440:]
    read = alt[275:525]
    rec, cig = check_pair(read, hap)
    assert rec[0] == 1 and rec[1] == 250 - (12 + 3 * 40) and ("D", 40) in cig


# ---- SPOA convex model ------------------------------------------------------------------------------------------

def poa_dump(seqs, sc=(0, -6, -6, -2, -26, -1)):
    lib = oracle()
    blob = b"".join(s + b"\0" for s in seqs)
    cap = 64 * sum(len(s) for s in seqs) + 1024
    out = (C.c_int32 * cap)()
    n = lib.orc_poa_align_dump(blob, len(seqs), *sc, out, cap)
    assert n > 0
    w = list(out[:n])
    V = w[0]
    pos = 1
    base, preds = [], []
    for _ in range(V):
        base.append(w[pos])
        k = w[pos + 1]
        preds.append(w[pos + 2: pos + 2 + k])
        pos += 2 + k
    order = w[pos:pos + V]
    pos += V
    A = w[pos]
    pairs = [(w[pos + 1 + 2 * x], w[pos + 2 + 2 * x]) for x in range(A)]
    return base, preds, order, pairs, w[pos + 1 + 2 * A]


def poa_alignment_score(seq, base, preds, pairs, gaps):
    """score of an alignment [(node | -1, seq pos | -1)]: the path must walk graph edges from a source to a sink and
    consume the sequence in order; a run of one gap kind costs the best affine model for its length"""
    def gap(L):
        return max(g + (L - 1) * e for g, e in gaps)
    succ_of = {v: set() for v in range(len(base))}
    for v, ps in enumerate(preds):
        for p in ps:
            succ_of[p].add(v)
    score, last_node, next_pos = 0, None, 0
    run_kind, run_len = None, 0

    def close():
        nonlocal score, run_kind, run_len
        if run_kind is not None:
            score += gap(run_len)
        run_kind, run_len = None, 0

    for node, pos in pairs:
        if node >= 0:
            if last_node is None:
                assert not preds[node], "alignment must start at a source node"
            else:
                assert node in succ_of[last_node], "alignment path leaves the graph"
            last_node = node
        if pos >= 0:
            assert pos == next_pos
            next_pos += 1
        if node >= 0 and pos >= 0:
            close()
            score += 0 if base[node] == seq[pos] else -6
        else:
            kind = "ins" if node < 0 else "del"
            if run_kind != kind:
                close()
                run_kind = kind
            run_len += 1
    close()
    assert next_pos == len(seq)
    assert last_node is not None and not succ_of[last_node], "alignment must end at a sink node"
    return score


@pytest.mark.parametrize("nh", [2, 3, 4, 5])
def test_poa_convex_alignment_is_optimal(nh):
    """production parameters (msa_builder.h:72-77): the optimum the restated SPOA engine computes for the last
    haplotype equals the optimum of an independent sequence-to-DAG DP with gap(L) =
    max(-6 - 2 (L-1), -26 - (L-1)) -- including 20/21/22-base gaps, where the two affine models cross."""
    rng = np.random.default_rng(500 + nh)
    gaps = ((-6, -2), (-26, -1))
    exact = 0
    for _ in range(12):
        haps = make_haplotypes(rng, nh, int(rng.integers(150, 400)))
        base, preds, order, pairs, dp_score = poa_dump(haps)
        sinks = [v for v in range(len(base)) if not any(v in p for p in preds)]
        want = bf.dag_global_score(haps[-1], np.array(base), preds, order, sinks)
        assert dp_score == want, (dp_score, want)
        # SPOA's backtrack follows "E or Q extends" jointly (restated literally in oracle/poa.cpp), so the path it
        # returns is a valid source-to-sink alignment but may be worth less than the optimum it started from
        got = poa_alignment_score(haps[-1], base, preds, pairs, gaps)
        assert got <= want
        exact += got == want
    assert exact >= 8


def test_poa_gap_crossover_is_at_21():
    """msa_builder.h:64 says the models cross "at exactly 20 bp" (6 + 2L = 26 + L); in SPOA's g + (L-1) e form
    (msa_builder.h:70) both cost 44 at L = 20 + ... : -6 - 2 (L-1) = -26 - (L-1) <=> L = 21.  Pinned: a lone 21-base
    deletion costs 46 under either model, 20 bases cost 44 (first model), 22 bases cost 47 (second model)."""
    assert bf.convex_gap(20) == -44 and bf.convex_gap(21) == -46 and bf.convex_gap(22) == -47
    rng = np.random.default_rng(3)
    for L, cost in ((20, -44), (21, -46), (22, -47), (40, -65)):
        ref = rand_dna(rng, 300)
        alt = ref[:150] + ref[150 + L:]
        base, preds, order, pairs, dp_score = poa_dump([ref, alt])
        got = poa_alignment_score(alt, base, preds, pairs, ((-6, -2), (-26, -1)))
        assert got == cost and dp_score == cost, (L, got, dp_score)


# ---- the closed forms poa.hip uses instead of a DP fill (k_msa: "alignments that need no fill") ----------------------

def _closed_form_pairs(ref, alt):
    """what poa.hip writes down for a first haplotype `alt` against the linear graph of `ref`, or None when it would
    run the DP: (a) equal length, <= 2 substitutions: the diagonal; (b) one indel and nothing else: shifted diagonal,
    the gap at its LEFTMOST place, main diagonal.  Pairs in path order, (node | -1, position | -1)."""
    V, L = len(ref), len(alt)
    if V == L:
        if sum(a != b for a, b in zip(ref, alt)) > 2:
            return None
        return [(i, i) for i in range(L)]
    m = min(V, L)
    lcp = next((i for i in range(m) if ref[i] != alt[i]), m)
    lcs = next((i for i in range(m) if ref[V - 1 - i] != alt[L - 1 - i]), m)
    if lcp + lcs < m:
        return None
    gl, a0 = abs(V - L), max(0, m - lcs)
    pairs = [(i, i) for i in range(a0)]
    if V > L:
        pairs += [(a0 + t, -1) for t in range(gl)] + [(a0 + gl + t, a0 + t) for t in range(L - a0)]
    else:
        pairs += [(-1, a0 + t) for t in range(gl)] + [(a0 + t, a0 + gl + t) for t in range(V - a0)]
    return pairs


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_poa_closed_forms_equal_the_restated_spoa_backtrack(seed):
    """The GPU path aligns a component's first haplotype without a fill when it is the backbone with <= 2 substitutions
    or with one indel and nothing else (proof in poa.hip).  Here the SAME closed form, restated in Python, is compared with
    the alignment the oracle's SPOA restatement returns -- substitutions anywhere (ends included), indels of 1 ... 40
    bases (across the 21-base crossover of the two gap models) in random sequence, in homopolymers and in tandem
    repeats, where the gap can slide and must land leftmost."""
    rng = np.random.default_rng(9100 + seed)
    checked = dict(sub=0, indel=0, slid=0)
    for trial in range(60):
        n = int(rng.integers(120, 320))
        ref = bytearray(rand_dna(rng, n))
        kind = trial % 4
        if kind == 1:  # a homopolymer run
            p = int(rng.integers(10, n - 60))
            ref[p:p + 25] = bytes([ref[p]]) * 25
        elif kind == 2:  # a tandem repeat of a 2-6 base unit
            u = rand_dna(rng, int(rng.integers(2, 7)))
            p = int(rng.integers(10, n - 80))
            ref[p:p + 48] = (u * 48)[:48]
        ref = bytes(ref)
        if kind == 3 or rng.random() < 0.25:  # substitutions
            alt = bytearray(ref)
            for q in rng.choice(n, size=int(rng.integers(1, 3)), replace=False):
                alt[q] = ord("A") if alt[q] != ord("A") else ord("C")
            if rng.random() < 0.3:
                alt[0 if rng.random() < 0.5 else n - 1] = ord("G") if ref[0] != ord("G") else ord("T")
            alt = bytes(alt)
            if sum(a != b for a, b in zip(ref, alt)) > 2:
                continue
            checked["sub"] += 1
        else:  # one indel, placed inside the repeat when there is one
            gl = int(rng.choice([1, 1, 2, 3, 5, 7, 12, 20, 21, 22, 30, 40]))
            at = int(rng.integers(0, n - gl - 1)) if kind == 0 else int(rng.integers(p, p + 20))
            if rng.random() < 0.5:
                alt = ref[:at] + ref[at + gl:]
            else:
                ins = ref[at:at + gl] if kind else rand_dna(rng, gl)  # a copy of the repeat slides, random bases do not
                alt = ref[:at] + ins + ref[at:]
            checked["indel"] += 1
        want = _closed_form_pairs(ref, alt)
        if want is None:
            continue  # (an inserted random base that happens to extend a run can make it two events: the DP's business)
        base, preds, order, pairs, dp_score = poa_dump([ref, alt])
        assert pairs == want, (trial, kind, len(ref), len(alt), pairs[:8], want[:8])
        if len(ref) != len(alt):
            m = min(len(ref), len(alt))
            lcp = next((i for i in range(m) if ref[i] != alt[i]), m)
            checked["slid"] += lcp > max(0, m - next((i for i in range(m) if ref[-1 - i] != alt[-1 - i]), m))
    assert checked["sub"] >= 10 and checked["indel"] >= 20 and checked["slid"] >= 5, checked


def _closed_form_later(base, preds, order, Q):
    """The closed form DESIGN.md section 9 works out for LATER haplotypes (measured on the GPU and not kept: the rounds of
    the POA stage are bound by a fill's latency, not by how many fills they hold): Q on a position-structured graph, best
    gapless path with <= 1 mismatch; None otherwise.  Explicit gapless scores Hd and the backtrack's rule (first predecessor
    in in-edge order with Hd(pred) + score == Hd(node); end node = first maximum among the sinks in rank order)."""
    V, n = len(base), len(Q)
    succ = [[] for _ in range(V)]
    for v, ps in enumerate(preds):
        for p in ps:
            succ[p].append(v)
    depth, Hd = [-1] * V, [0] * V
    for v in order:
        ds = {depth[p] for p in preds[v]}
        if len(ds) > 1:
            return None  # not position-structured
        d = ds.pop() + 1 if ds else 0
        if d >= n:
            return None
        depth[v] = d
        Hd[v] = (0 if base[v] == Q[d] else -6) + (max(Hd[p] for p in preds[v]) if preds[v] else 0)
    sinks = [v for v in order if not succ[v]]
    if any(depth[v] != n - 1 for v in sinks):
        return None
    best = sinks[0]
    for v in sinks:
        if Hd[v] > Hd[best]:
            best = v
    if Hd[best] < -6:
        return None
    path, cur = [], best
    while True:
        path.append((cur, depth[cur]))
        if not preds[cur]:
            break
        s = 0 if base[cur] == Q[depth[cur]] else -6
        cur = next(p for p in preds[cur] if Hd[p] + s == Hd[cur])
    return path[::-1]


@pytest.mark.parametrize("seed", [0, 1])
def test_poa_closed_form_for_later_haplotypes_equals_the_restated_spoa_backtrack(seed):
    """graphs of 2-5 same-length haplotypes that differ by substitutions at shared, tri-allelic and adjacent sites; the
    last haplotype's alignment by the gapless closed form (<= 1 mismatch against its best path) equals the oracle's"""
    rng = np.random.default_rng(9300 + seed)
    applied = with_mismatch = 0
    for trial in range(150):
        n = int(rng.integers(60, 160))
        ref = bytearray(rand_dna(rng, n))
        sites = sorted(rng.choice(n, size=int(rng.integers(1, 6)), replace=False).tolist())
        if rng.random() < 0.4 and len(sites) >= 2 and sites[0] + 1 < n:
            sites[1] = sites[0] + 1
        if rng.random() < 0.3:
            sites[0] = 0 if rng.random() < 0.5 else n - 1
        alt1 = {q: (ord("A") if ref[q] != ord("A") else ord("C")) for q in sites}
        alt2 = {q: next(c for c in b"GTCA" if c != ref[q] and c != alt1[q]) for q in sites}
        haps = [bytes(ref)]
        for _ in range(int(rng.integers(2, 5))):
            a = bytearray(ref)
            for q in sites:
                r = rng.random()
                if r < 0.35:
                    a[q] = alt1[q]
                elif r < 0.5:
                    a[q] = alt2[q]
            if bytes(a) not in haps:
                haps.append(bytes(a))
        if len(haps) < 3:
            continue
        base, preds, order, pairs, dp_score = poa_dump(haps)
        want = _closed_form_later(base, preds, order, haps[-1])
        if want is None:
            continue
        applied += 1
        with_mismatch += dp_score == -6
        assert pairs == want, (trial, pairs[:6], want[:6])
        assert dp_score in (0, -6)
    assert applied >= 60 and with_mismatch >= 15, (applied, with_mismatch)
