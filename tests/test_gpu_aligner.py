"""GPU parity of the read<->haplotype aligner on hand-made haplotypes: every route a pair can take -- the three
gapless certificates, the six register width classes, the wavefront-per-pair kernel, the last-resort HBM-row kernel -- against
the oracle (whose scores tests/test_aligner_pins.py pins to a brute force).  The assembly and variant buffers are
written by hand, so the cases do not depend on what the assembler makes of a window."""
import numpy as np
import pytest

from harness import OracleEngine, compare_geno, handmade_annotation_case
from lancet2_amd import capi, synth
from pin_cases import BASES, CASES, make_pair, mutate, rand_dna

pytestmark = pytest.mark.gpu
SWEEP = int(__import__("os").environ.get("MA_SWEEP_SEED", "0"))  # tools: other seeded cases of the same shapes


def _read(seq, qname, rev=False, sample=0, role=0):
    q = np.full(len(seq), 30, np.uint8)
    return dict(seq=np.frombuffer(seq, np.uint8).copy(), qual=q, qname=qname, sample=sample, role=role, rev=rev,
                passf=True, start=0, hint=capi.MA_NO_HINT)


def _window(haps, reads):
    """one window: haplotypes[0] is the REF haplotype; one SNV-like variant in the middle so that it is genotyped"""
    pos = min(len(h) for h in haps) // 2
    case = dict(haps=[h.decode() for h in haps], ref_pos=pos, ref_len=1,
                alts=[(1, {h: pos for h in range(1, len(haps))})] if len(haps) > 1 else [(1, {})])
    win = dict(ref=np.frombuffer(haps[0][:600].ljust(64, b"A"), np.uint8).copy(),
               reads=[_read(r, i // 2, rev=bool(i & 1), sample=i % 2, role=i % 2) for i, r in enumerate(reads)])
    return case, win


def _run(params, cases, wins):
    asm, var = handmade_annotation_case(params, cases)
    arrs, n, nr = synth.pack_batch(wins)
    orc = OracleEngine(params)
    want = orc.genotype(arrs, n, nr, asm, var)
    from lancet2_amd.engine import Engine
    out = {}
    for tier in (0, 1, 2, 3):
        p = capi.default_params(**{f: getattr(params, f) for f, _ in capi.Params._fields_})
        p.aln_tier = tier
        eng = Engine(p)
        try:
            got = eng.genotype(arrs, n, nr, asm, var)
            if tier == 0:
                out = dict(eng.kernel_times())
        finally:
            eng.close()
        bad = compare_geno(params, got, want, n, nr, var["win_nvars"], arrs["read_win_off"])
        assert not bad, f"aln_tier {tier}:\n" + "\n".join(bad[:20])
    return want, out


def test_every_pin_case_flavour():
    """the flavours of tests/pin_cases.py (clean, noisy, indels up to 40, STR, overhangs, N, unrelated), 40 reads per
    haplotype pair"""
    params = capi.default_params(min_k=25, max_k=25, max_hap_len=2048)
    rng = np.random.default_rng(4242 + SWEEP)
    cases, wins = [], []
    for case in CASES:
        for _ in range(3):
            read0, hap = make_pair(rng, case)
            alt = mutate(rng, hap, sub=0.002, indels=[("D", 7)])
            reads = [read0]
            for _ in range(39):
                r, _h = make_pair(rng, case)
                # re-anchor the read on THIS haplotype: a mutated substring (or an overhanging / unrelated one)
                m = len(r)
                if case == "unrelated":
                    reads.append(r)
                    continue
                st = int(rng.integers(-40, max(len(hap) - m + 40, -39)))
                src = rand_dna(rng, 60) + hap + rand_dna(rng, 60)
                sub = src[60 + st: 60 + st + m + 30]
                k = int(rng.integers(0, 3))
                sub = mutate(rng, sub, sub=float(rng.choice([0.0, 0.01, 0.04])),
                             indels=[(str(rng.choice(["I", "D"])), int(rng.choice([1, 2, 4, 9, 15, 22, 35]))) for _ in range(k)])
                if case == "amb" and len(sub) > 20:
                    b = bytearray(sub)
                    b[int(rng.integers(0, len(b)))] = ord("N")
                    sub = bytes(b)
                reads.append(sub[:m])
            c, w = _window([hap, alt], reads)
            cases.append(c)
            wins.append(w)
    want, _ = _run(params, cases, wins)
    rec = want["aln_rec"].reshape(-1, 6)
    assert (rec[:, 0] > 0).sum() > 600


@pytest.mark.parametrize("wave_max", [None, "512"])
def test_wide_regions_tandem_repeats_and_duplications(wave_max, monkeypatch):
    """seeds spread over many diagonals: tandem repeats (tens to ~200 diagonals: upper register classes and the
    wavefront-per-pair kernel) and a 300-base segment duplicated 650 bases downstream (two anchor diagonals 650 apart:
    the big wavefront class, or -- with the LDS row capped by MA_WAVE_MAX_W -- the last-resort kernel with its row
    in HBM)"""
    if wave_max:
        monkeypatch.setenv("MA_WAVE_MAX_W", wave_max)
    params = capi.default_params(min_k=25, max_k=25, max_hap_len=2048)
    rng = np.random.default_rng(77 + SWEEP)
    cases, wins = [], []
    for unit_len, rep_len in ((1, 40), (2, 60), (3, 90), (5, 150), (6, 240)):
        hap = rand_dna(rng, 1200)
        rep = (rand_dna(rng, unit_len) * 300)[:rep_len]
        hap = hap[:500] + rep + hap[500 + rep_len:]
        alt = hap[:500] + rep[: rep_len - 2 * unit_len] + hap[500 + rep_len:]   # two units shorter
        reads = []
        for _ in range(48):
            st = int(rng.integers(330, 500 + rep_len - 20))
            src = hap if rng.random() < 0.5 else alt
            r = src[st: st + int(rng.choice([101, 150, 250]))]
            reads.append(mutate(rng, r, sub=float(rng.choice([0.0, 0.01, 0.03]))))
        c, w = _window([hap, alt], reads)
        cases.append(c)
        wins.append(w)
    hap = rand_dna(rng, 1500)
    hap = hap[:900] + hap[250:550] + hap[1200:]     # hap[250:550] again at 900
    alt = hap[:400] + hap[410:]
    reads = []
    for _ in range(48):
        st = int(rng.integers(150, 1250))
        reads.append(mutate(rng, hap[st: st + 150], sub=float(rng.choice([0.0, 0.01]))))
    c, w = _window([hap, alt], reads)
    cases.append(c)
    wins.append(w)
    want, kernels = _run(params, cases, wins)
    assert "k_align_wave" in kernels and (("k_align_gen" in kernels) == bool(wave_max)), kernels
    assert (want["aln_rec"].reshape(-1, 6)[:, 0] > 0).sum() > 300


def test_long_reads_and_long_indels():
    """250-400 base reads (K = 52 .. 102: classes 97 / 129 / LDS) across 20-60 base indels, and 150-base reads across
    100-300 base deletions: no hit end to end against the haplotype without the deletion (cost 12 + 3 L), a clean
    hit against the one that carries it"""
    params = capi.default_params(min_k=25, max_k=25, max_hap_len=2048)
    rng = np.random.default_rng(31 + SWEEP)
    cases, wins = [], []
    for m in (250, 300, 400):
        hap = rand_dna(rng, 1400)
        alt = hap[:600] + hap[600 + 45:]
        alt2 = hap[:800] + rand_dna(rng, 30) + hap[800:]
        reads = []
        for _ in range(40):
            src = [hap, alt, alt2][int(rng.integers(0, 3))]
            st = int(rng.integers(300, 900))
            reads.append(mutate(rng, src[st: st + m], sub=float(rng.choice([0.0, 0.01, 0.02]))))
        c, w = _window([hap, alt, alt2], reads)
        cases.append(c)
        wins.append(w)
    for L in (100, 200, 300):
        hap = rand_dna(rng, 1600)
        alt = hap[:700] + hap[700 + L:]
        reads = [alt[700 - 75 - d: 700 + 75 - d] for d in range(-60, 61, 6)] + \
                [hap[st: st + 150] for st in range(500, 1100, 40)]
        c, w = _window([hap, alt], reads)
        cases.append(c)
        wins.append(w)
    want, _ = _run(params, cases, wins)
    rec = want["aln_rec"].reshape(-1, 6)
    assert (rec[:, 0] > 0).sum() > 150 and (rec[:, 0] == 0).sum() > 20


def test_pairs_whose_dp_region_the_vote_narrows():
    """Round 5: k_vote narrows a DP pair's region from a lower bound of its optimum (align.hip: vote_settle) -- the gapless
    path on the most-voted diagonal, a one-gap path between the two most-voted diagonals, minus the rows no path can pair,
    reads with N -- and the kernels fill only the chunks of a row that the widest region of a group reaches.  The flavours
    that take each of those routes, and the ones that must NOT be narrowed (soft-clipped ends: an optimum below what the
    anchor argument needs), against the oracle's full region; tiers 2 / 3 of _run are the same engine without the
    narrowing."""
    params = capi.default_params(min_k=25, max_k=25, max_hap_len=2048)
    rng = np.random.default_rng(909 + SWEEP)
    cases, wins = [], []
    for m in (150, 101, 250):
        hap = rand_dna(rng, 1100)
        alt1 = hap[:500] + hap[502:]                      # 2-base deletion: two diagonals 2 apart
        alt2 = hap[:640] + rand_dna(rng, 1) + hap[640:]   # 1-base insertion
        alt3 = hap[:560] + hap[567:]                      # 7-base deletion
        reads = []
        for src, site in ((alt1, 500), (alt2, 640), (alt3, 560)):
            for off in range(6, m - 4, max(m // 12, 1)):  # the indel 6 bases from the read's start ... 4 from its end
                reads.append(mutate(rng, src[site - off: site - off + m], sub=float(rng.choice([0.0, 0.0, 0.01]))))
        for x in (3, 4, 6, 9, 12, 15):                    # mismatches only: S0 = m - 5 x is the bound
            st = int(rng.integers(50, len(hap) - m - 1))
            b = bytearray(hap[st: st + m])
            for p in rng.choice(m, size=x, replace=False):
                b[p] = BASES[(BASES.index(bytes([b[p]])) + 1) % 4]
            reads.append(bytes(b))
        for o in (52, 58, 63, 64, 66, 69, 70, 71, 75):   # hanging over an end by o bases, with and without a mismatch
            if o >= m - 30:
                continue
            left = rand_dna(rng, o) + hap[: m - o]
            right = hap[len(hap) - (m - o):] + rand_dna(rng, o)
            reads += [left, right, mutate(rng, left, sub=0.01), mutate(rng, right, sub=0.01)]
        for _ in range(8):                                # an N and an indel / a few mismatches
            st = int(rng.integers(380, 520))
            b = bytearray(mutate(rng, alt1[st: st + m], sub=0.01))
            b[int(rng.integers(0, len(b)))] = ord("N")
            reads.append(bytes(b))
        for clip in (8, 14, 20, 26, 34):                  # soft-clipped ends: adapter-like bases the mapper clipped
            st = int(rng.integers(50, len(hap) - m - 1))
            reads.append(hap[st: st + m - clip] + rand_dna(rng, clip))
            reads.append(rand_dna(rng, clip) + hap[st + clip: st + m])
        c, w = _window([hap, alt1, alt2, alt3], reads)
        cases.append(c)
        wins.append(w)
    want, kt = _run(params, cases, wins)
    rec = want["aln_rec"].reshape(-1, 6)
    assert (rec[:, 0] > 0).sum() > 400 and (rec[:, 0] == 0).sum() > 20
    assert kt.get("k_align_reg", 0.0) > 0.0


def test_more_than_4096_pairs_in_the_65_cell_class():
    """width class 3 (regions of 50-65 cells) shares reg_waves() == 3 with the classes of <= 49 cells, and the
    packed / paired launch took it for one of them once it held more than 4096 DP pairs (below that the class is rerouted to the
    wavefront kernel): its pairs ran through a 49-cell row body.  180-base reads have K = 29, i.e. regions of 59 + spread cells
    whenever the region is not narrowed: reads whose last 80 bases are foreign (an optimum of ~100 -- too poor for the anchor
    argument, so tier 0 keeps the full region) and, with the certificates off (tier 2), every read."""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25, max_hap_len=2048)
    rng = np.random.default_rng(6565 + SWEEP)
    cases, wins = [], []
    for _ in range(16):
        hap = rand_dna(rng, 900)
        alt = hap[:450] + hap[453:]
        reads = []
        for i in range(300):
            st = int(rng.integers(0, len(hap) - 181))
            src = hap if i & 1 else alt
            if i % 3 == 0:
                reads.append(src[st: st + 100] + rand_dna(rng, 80))
            elif i % 3 == 1:
                reads.append(rand_dna(rng, 80) + src[st + 80: st + 180])
            else:
                reads.append(mutate(rng, src[st: st + 180], sub=0.03))
        c, w = _window([hap, alt], reads)
        cases.append(c)
        wins.append(w)
    asm, var = handmade_annotation_case(params, cases)
    arrs, n, nr = synth.pack_batch(wins)
    want = OracleEngine(params).genotype(arrs, n, nr, asm, var)
    for tier in (0, 2):
        p = capi.default_params(**{f: getattr(params, f) for f, _ in capi.Params._fields_})
        p.aln_tier = tier
        eng = Engine(p)
        try:
            got = eng.genotype(arrs, n, nr, asm, var)
            st = eng.stats()
            kt = dict(eng.kernel_times())
        finally:
            eng.close()
        assert st["dp_w65"] > 4096 and "k_align_reg" in kt, (tier, st, kt)
        bad = compare_geno(params, got, want, n, nr, var["win_nvars"], arrs["read_win_off"])
        assert not bad, f"aln_tier {tier}:\n" + "\n".join(bad[:20])
