"""GPU parity tests: the HIP path (through the C-ABI) against the oracle on identical seeded inputs.
Bit-exact for every integer/byte/index output; f64 statistics compared bit-for-bit as well (both sides
use the same sequential IEEE operations with FMA contraction off)."""
import numpy as np
import pytest

from harness import OracleEngine
from lancet2_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from lancet2_amd.engine import Engine
    e = Engine(capi.default_params())
    yield e
    e.close()


def _planted_repeat_batch():
    wins = []
    rng = np.random.default_rng(11)
    for i in range(6):
        w = synth.make_window(100 + i, W=1001, depths=(2, 2))
        ref = w["ref"]
        L = [0, 20, 30, 61, 130, 300][i]
        if L:
            ref[600:600 + L] = ref[100:100 + L]
            if i >= 3:  # sprinkle two mismatches inside the copy
                ref[600 + L // 3] = ord("A") if ref[600 + L // 3] != ord("A") else ord("C")
                ref[600 + 2 * L // 3] = ord("G") if ref[600 + 2 * L // 3] != ord("G") else ord("T")
        wins.append(w)
    wins.append(dict(ref=np.frombuffer(b"ACGT" * 50, dtype=np.uint8).copy(), reads=[]))     # pure STR
    wins.append(dict(ref=np.frombuffer(b"N" * 300, dtype=np.uint8).copy(), reads=[]))       # all N
    wins.append(dict(ref=np.frombuffer(b"ACGTTGCA", dtype=np.uint8).copy(), reads=[]))      # tiny
    wins.append(dict(ref=np.frombuffer(b"A", dtype=np.uint8).copy(), reads=[]))             # 1 base
    return synth.pack_batch(wins)


def test_repeat_gate_parity(engine):
    arrs, n, nr = _planted_repeat_batch()
    want = OracleEngine(engine.p).gate(arrs, n, nr)
    got = engine.gate(arrs, n, nr)
    assert np.array_equal(got["max_approx"], want["max_approx"]), (got["max_approx"], want["max_approx"])
    assert np.array_equal(got["max_exact"], want["max_exact"])


def test_repeat_gate_parity_random(engine):
    arrs, n, nr = synth.make_config_batch("C1", 8, depths=(1, 1))
    want = OracleEngine(engine.p).gate(arrs, n, nr)
    got = engine.gate(arrs, n, nr)
    assert np.array_equal(got["max_approx"], want["max_approx"])
    assert np.array_equal(got["max_exact"], want["max_exact"])


@pytest.mark.parametrize("mm", [0, 1, 2, 3])
def test_repeat_gate_parity_window_lengths(mm):
    """window lengths around the kernel's word loop (len % 4), its shifted-copy limit (2560) and its maximum (8192),
    with planted approximate repeats, for every mismatch budget"""
    from lancet2_amd.engine import Engine
    rng = np.random.default_rng(77 + mm)
    wins = []
    for W in (5, 6, 7, 8, 97, 1001, 2499, 2560, 2561, 3001, 8192):
        ref = synth.BASES[rng.integers(0, 4, W)].copy()
        if W > 200:  # an approximate repeat of 60 bases with `mm` substitutions, far apart
            src, dst = 10, W - 80
            ref[dst:dst + 60] = ref[src:src + 60]
            for x in rng.choice(60, size=mm, replace=False):
                ref[dst + x] = ord("A") if ref[dst + x] != ord("A") else ord("C")
        wins.append(dict(ref=ref, reads=[]))
    arrs, n, nr = synth.pack_batch(wins)
    params = capi.default_params(max_mismatch=mm)
    want = OracleEngine(params).gate(arrs, n, nr)
    eng = Engine(params)
    try:
        got = eng.gate(arrs, n, nr)
    finally:
        eng.close()
    assert np.array_equal(got["max_approx"], want["max_approx"]), (got["max_approx"], want["max_approx"])
    assert np.array_equal(got["max_exact"], want["max_exact"])
    assert int(want["max_approx"][-1]) >= 60


@pytest.mark.parametrize("mm", [0, 1, 2, 3])
def test_repeat_gate_two_diagonals_per_lane(mm):
    """Round 6: the gate walks TWO diagonals per lane in packed 16-bit halves (diagonal d + 1 is one position shorter than d
    and is fed mismatches past its end).  Every window length from 2 to 72 and around the workgroup's strides (256 lanes x 2
    diagonals), periodic sequences (every diagonal a long run), repeats that end on the window's last base, homopolymers, and
    bytes that are not upper-case ACGT (N, lower case: the gate compares bytes) -- equal to the oracle for every budget."""
    from lancet2_amd.engine import Engine
    rng = np.random.default_rng(4100 + mm)
    wins = []
    for W in list(range(2, 73)) + [255, 256, 257, 511, 512, 513, 514, 1023, 1024, 1025, 1026, 2559]:
        kind = W % 5
        if kind == 0:
            ref = np.frombuffer((b"ACGGT" * (W // 5 + 1))[:W], dtype=np.uint8).copy()          # period 5
        elif kind == 1:
            ref = np.full(W, ord("A"), dtype=np.uint8)                                          # homopolymer
        else:
            ref = synth.BASES[rng.integers(0, 4, W)].copy()
            if W >= 24:  # a repeat whose second copy ends on the window's last base, with `mm` substitutions
                L = min(W // 2 - 1, 40)
                ref[W - L:] = ref[1:1 + L]
                for x in rng.choice(L, size=min(mm, L), replace=False):
                    ref[W - L + x] = ord("A") if ref[W - L + x] != ord("A") else ord("C")
            if kind == 3 and W >= 8:
                ref[rng.integers(0, W, 3)] = ord("N")
            if kind == 4 and W >= 8:
                idx = rng.integers(0, W, 4)
                ref[idx] = ref[idx] | 0x20  # lower case
        wins.append(dict(ref=ref, reads=[]))
    arrs, n, nr = synth.pack_batch(wins)
    params = capi.default_params(max_mismatch=mm)
    want = OracleEngine(params).gate(arrs, n, nr)
    eng = Engine(params)
    try:
        got = eng.gate(arrs, n, nr)
    finally:
        eng.close()
    assert np.array_equal(got["max_approx"], want["max_approx"]), [(i, int(a), int(b)) for i, (a, b) in enumerate(zip(got["max_approx"], want["max_approx"])) if a != b][:10]
    assert np.array_equal(got["max_exact"], want["max_exact"]), [(i, int(a), int(b)) for i, (a, b) in enumerate(zip(got["max_exact"], want["max_exact"])) if a != b][:10]


from harness import compare_asm  # noqa: E402


def _asm_parity(params, arrs, n, nr):
    from lancet2_amd.engine import Engine
    eng = Engine(params)
    try:
        got = eng.assemble(arrs, n, nr)
    finally:
        eng.close()
    want = OracleEngine(params).assemble(arrs, n, nr)
    bad = compare_asm(params, got, want, n)
    assert not bad, "\n".join(bad[:20])
    return got, want


@pytest.mark.parametrize("cfg,nwin", [("C1", 4), ("C2", 6), ("C3", 4)])
def test_assemble_parity_fixed_k25(cfg, nwin):
    arrs, n, nr = synth.make_config_batch(cfg, nwin)
    got, want = _asm_parity(capi.default_params(min_k=25, max_k=25), arrs, n, nr)
    assert (want["win_ncomp"] > 0).sum() >= nwin // 2  # the synthetic windows do assemble


def test_assemble_parity_k_cascade():
    arrs, n, nr = synth.make_config_batch("C2", 6, first_index=40)
    _asm_parity(capi.default_params(), arrs, n, nr)


def test_assemble_parity_three_samples():
    arrs, n, nr = synth.make_config_batch("C5", 4, first_index=80)
    _asm_parity(capi.default_params(min_k=25, max_k=25, num_samples=3), arrs, n, nr)


def test_assemble_parity_str_and_edge_cases():
    wins = [synth.make_window(200 + i, W=1001, depths=(30, 30), str_unit=u) for i, u in enumerate([b"CA", b"AAG", b"T"])]
    wins.append(synth.make_window(210, W=1001, depths=(0, 0)))               # no reads at all
    wins.append(synth.make_window(211, W=1001, depths=(3, 3)))               # below anchor coverage
    wins.append(dict(ref=np.frombuffer(b"N" * 400, dtype=np.uint8).copy(), reads=[]))
    wins.append(synth.make_window(212, W=300, depths=(30, 30)))              # short window
    arrs, n, nr = synth.pack_batch(wins)
    _asm_parity(capi.default_params(), arrs, n, nr)


from harness import compare_vars  # noqa: E402


@pytest.mark.parametrize("cfg,nwin,kw", [("C2", 8, {}), ("C3", 4, {}), ("C4", 1, dict(depths=(60, 60))),
                                          ("C5", 3, dict(num_samples=3)),
                                          # haplotypes longer than 1024 / 1536 bases: 6- and 8-column lanes of the POA fill
                                          ("C2", 2, dict(W=1300)), ("C2", 2, dict(W=1700)),
                                          # a big insertion / deletion: long left / up extension runs in the traceback
                                          ("C2", 3, dict(big_indel=60)), ("C2", 3, dict(big_indel=35, depths=(50, 50)))])
def test_msa_parity(cfg, nwin, kw):
    from lancet2_amd.engine import Engine
    ns = kw.pop("num_samples", 2)
    params = capi.default_params(min_k=25, max_k=25, num_samples=ns)
    arrs, n, nr = synth.make_config_batch(cfg, nwin, first_index=300, **kw)
    orc = OracleEngine(params)
    asm = orc.assemble(arrs, n, nr)
    want = orc.msa(arrs, n, nr, asm)
    eng = Engine(params)
    try:
        got = eng.msa(arrs, n, nr, asm)
    finally:
        eng.close()
    bad = compare_vars(params, got, want, n)
    assert not bad, "\n".join(bad[:20])
    assert want["win_nvars"].sum() > 0


@pytest.mark.parametrize("band_mode", ["0", "1"])
@pytest.mark.parametrize("cfg,nwin,kw", [("C2", 6, {}), ("C2", 2, dict(W=1700)), ("C2", 3, dict(big_indel=60))])
def test_msa_parity_fill_modes(cfg, nwin, kw, band_mode, monkeypatch):
    """The default is the banded POA fill in its own kernel (MA_POA_BAND=2, exercised by every other test); the
    full row-synchronous fill (0) and the banded fill inside k_msa (1) must give the same bits."""
    from lancet2_amd.engine import Engine
    monkeypatch.setenv("MA_POA_BAND", band_mode)
    params = capi.default_params(min_k=25, max_k=25)
    arrs, n, nr = synth.make_config_batch(cfg, nwin, first_index=300, **kw)
    orc = OracleEngine(params)
    asm = orc.assemble(arrs, n, nr)
    want = orc.msa(arrs, n, nr, asm)
    eng = Engine(params)
    try:
        got = eng.msa(arrs, n, nr, asm)
    finally:
        eng.close()
    bad = compare_vars(params, got, want, n)
    assert not bad, "\n".join(bad[:20])


@pytest.mark.parametrize("min_pending", ["0", "1000000", None])
@pytest.mark.parametrize("tier0", ["1", "2", "4"])
def test_msa_parity_band_tiers(tier0, min_pending, monkeypatch):
    """The banded fill starts with a narrow tier (64 or 128 columns per row, MA_POA_TIER0) and hands an alignment whose
    certificate fails to the 256-column tier, then to the full fill; while many windows wait, the fills run in k_msa_band
    rounds, the tail of a batch inside k_msa (MA_POA_MIN_PENDING = 0: always rounds, huge: never).  Same bits every way,
    on plain windows, variant-dense windows (several alignments per window) and 60-base indels (the narrow tiers fail)."""
    from lancet2_amd.engine import Engine
    monkeypatch.setenv("MA_POA_TIER0", tier0)
    monkeypatch.setenv("MA_POA_NO_DIRECT", "1")  # every alignment takes a fill
    if min_pending is not None:  # (round 6: the host-counted rounds are MA_POA_SCHED=0; None = the persistent kernel k_poa)
        monkeypatch.setenv("MA_POA_MIN_PENDING", min_pending)
        monkeypatch.setenv("MA_POA_SCHED", "0")
    params = capi.default_params(min_k=25, max_k=25)
    for cfg, nwin, kw in (("C2", 5, {}), ("C2", 3, dict(big_indel=60)), ("C2", 3, dict(snv_rate=8e-3, indel_rate=2e-3)),
                          ("C2", 2, dict(W=1700))):
        arrs, n, nr = synth.make_config_batch(cfg, nwin, first_index=5200, **kw)
        orc = OracleEngine(params)
        asm = orc.assemble(arrs, n, nr)
        want = orc.msa(arrs, n, nr, asm)
        eng = Engine(params)
        try:
            got = eng.msa(arrs, n, nr, asm)
        finally:
            eng.close()
        bad = compare_vars(params, got, want, n)
        assert not bad, (cfg, kw, "\n".join(bad[:20]))


@pytest.mark.parametrize("sched,wgs", [("1", None), ("1", "1"), ("0", None), ("1", "xcd0")])
def test_msa_persistent_kernel_on_a_batch_that_overfills_the_chip(sched, wgs, monkeypatch):
    """Round 6: k_poa schedules the windows on the device (fresh windows off a counter, fills and resumed windows through two
    queues in HBM, windows handed from workgroup to workgroup through their LDS image).  1536 windows -- three times what the
    chip holds at once, so workgroups take window after window and fills of many windows are in flight while others are
    resumed -- with every alignment through a fill (no closed forms) and indels that fail the first tier: the variants of all
    1536 equal the oracle's for the 96 distinct windows they are copies of; two workgroups per CU, one, the host-counted
    rounds (MA_POA_SCHED=0) and the single-domain variant with device-scope fences (MA_POA_XCD=0) alike."""
    from lancet2_amd.engine import Engine
    monkeypatch.setenv("MA_POA_SCHED", sched)
    monkeypatch.setenv("MA_POA_NO_DIRECT", "1")
    if wgs == "xcd0":  # one scheduling domain for the whole chip, device-scope fences around every hand-over (MA_POA_XCD=0)
        monkeypatch.setenv("MA_POA_XCD", "0")
    elif wgs:
        monkeypatch.setenv("MA_POA_WGS_PER_CU", wgs)
    params = capi.default_params(min_k=25, max_k=25)
    parts = [synth.make_config_batch("C2", 48, first_index=8100), synth.make_config_batch("C2", 24, first_index=8300, big_indel=60),
             synth.make_config_batch("C3", 24, first_index=8400, snv_rate=8e-3, indel_rate=2e-3)]
    import bench
    arrs0, n0, nr0 = bench.concat_batches(parts)
    orc = OracleEngine(params)
    asm0 = orc.assemble(arrs0, n0, nr0)
    want0 = orc.msa(arrs0, n0, nr0, asm0)
    times = 16
    arrs, n, nr = synth.tile_batch(arrs0, n0, nr0, times)
    asm = {k: np.tile(v, times) for k, v in asm0.items()}
    want = {k: np.tile(v, times) for k, v in want0.items()}
    eng = Engine(params)
    try:
        for _ in range(2):  # (a second call reuses the workspace: stale queue slots / images must not matter)
            got = eng.msa(arrs, n, nr, asm)
            bad = compare_vars(params, got, want, n)
            assert not bad, "\n".join(bad[:20])
        kt = dict(eng.kernel_times())
    finally:
        eng.close()
    assert ("k_poa" in kt) == (sched == "1"), kt
    assert want0["win_nvars"].sum() > 100


def test_msa_full_fill_inside_the_persistent_kernel(monkeypatch):
    """A window's own code / stored-row areas hold what the band tiers write (256 columns per row at most); the full
    row-synchronous fill -- the last tier -- takes whole-row areas that belong to the WORKGROUP of k_poa that runs it.
    MA_POA_BAND=3: k_poa with every alignment through the full fill, on 768 windows (more than the 512 workgroups: an area
    serves window after window) of three shapes -- the variants equal the oracle's."""
    from lancet2_amd.engine import Engine
    monkeypatch.setenv("MA_POA_BAND", "3")
    monkeypatch.setenv("MA_POA_NO_DIRECT", "1")
    params = capi.default_params(min_k=25, max_k=25)
    parts = [synth.make_config_batch("C2", 24, first_index=8600), synth.make_config_batch("C2", 12, first_index=8700, big_indel=60),
             synth.make_config_batch("C3", 12, first_index=8800, snv_rate=8e-3, indel_rate=2e-3)]
    import bench
    arrs0, n0, nr0 = bench.concat_batches(parts)
    orc = OracleEngine(params)
    asm0 = orc.assemble(arrs0, n0, nr0)
    want0 = orc.msa(arrs0, n0, nr0, asm0)
    times = 16
    arrs, n, nr = synth.tile_batch(arrs0, n0, nr0, times)
    asm = {k: np.tile(v, times) for k, v in asm0.items()}
    want = {k: np.tile(v, times) for k, v in want0.items()}
    eng = Engine(params)
    try:
        eng.timing_control(3)  # (collect the device-side work counters too)
        got = eng.msa(arrs, n, nr, asm)
        bad = compare_vars(params, got, want, n)
        assert not bad, "\n".join(bad[:20])
        kt = dict(eng.kernel_times())
        st = eng.stats()
    finally:
        eng.close()
    assert "k_poa" in kt, kt
    assert st["poa_full_cells"] > 0 and st["poa_band_fills"] == 0, st
    assert want0["win_nvars"].sum() > 50


@pytest.mark.parametrize("lean,raw_cap", [("0", None), ("1", "4"), ("1", "40")])
def test_msa_parity_lean_fill_and_bubble_scratch(lean, raw_cap, monkeypatch):
    """poa_fill_lean (default) against poa_fill_band (MA_POA_LEAN=0): same decision codes, so the same variants; and the
    bubble walk's raw alleles in LDS (default) against the HBM route a bubble takes when an allele outgrows its LDS share
    (MA_POA_RAW_CAP = 4 / 40 bytes: every bubble / the 60-base ones move)."""
    from lancet2_amd.engine import Engine
    monkeypatch.setenv("MA_POA_LEAN", lean)
    if raw_cap:
        monkeypatch.setenv("MA_POA_RAW_CAP", raw_cap)
    params = capi.default_params(min_k=25, max_k=25)
    for cfg, nwin, kw in (("C2", 6, {}), ("C2", 3, dict(big_indel=60)), ("C3", 3, dict(snv_rate=8e-3, indel_rate=2e-3))):
        arrs, n, nr = synth.make_config_batch(cfg, nwin, first_index=6400, **kw)
        orc = OracleEngine(params)
        asm = orc.assemble(arrs, n, nr)
        want = orc.msa(arrs, n, nr, asm)
        eng = Engine(params)
        try:
            got = eng.msa(arrs, n, nr, asm)
        finally:
            eng.close()
        bad = compare_vars(params, got, want, n)
        assert not bad, (cfg, kw, "\n".join(bad[:20]))
        assert want["win_nvars"].sum() > 0


@pytest.mark.parametrize("no_direct", [False, True])
@pytest.mark.parametrize("kw,need", [(dict(indel_rate=6e-4), (2, 2, 2)),
                                     (dict(str_unit=b"A", indel_rate=4e-4), (2, 4, 2))])  # homopolymers: the indel slides
def test_msa_parity_haplotypes_aligned_without_a_fill(kw, need, no_direct, monkeypatch):
    """A first haplotype that is the reference haplotype with one or two substitutions, or with one indel and nothing
    else, is aligned without a DP fill (SPOA's backtrack is known in closed form, poa.hip); with MA_POA_NO_DIRECT it takes
    the fill like any other: same bits, and the batch really holds such haplotypes of each kind (and others)."""
    from lancet2_amd.engine import Engine
    if no_direct:
        monkeypatch.setenv("MA_POA_NO_DIRECT", "1")
    params = capi.default_params(min_k=25, max_k=25)
    arrs, n, nr = synth.make_config_batch("C2", 24, first_index=4100, **kw)
    orc = OracleEngine(params)
    asm = orc.assemble(arrs, n, nr)
    want = orc.msa(arrs, n, nr, asm)
    MH, ML, MC = params.max_haps, params.max_hap_len, params.max_comps
    hl, hb = asm["hap_len"].reshape(n, MH), asm["hap_bases"].reshape(n, MH, ML)
    single = indel = other = 0
    for w in range(n):
        for c in range(int(asm["win_ncomp"][w])):
            b, k = int(asm["comp_hap0"].reshape(n, MC)[w, c]), int(asm["comp_nhaps"].reshape(n, MC)[w, c])
            if k < 2:
                continue
            R, Q = hb[w, b, :hl[w, b]], hb[w, b + 1, :hl[w, b + 1]]
            m = min(len(R), len(Q))
            if len(R) == len(Q):
                d = int((R != Q).sum())
                single += d <= 2
                other += d > 2
            else:
                neq = np.nonzero(R[:m] != Q[:m])[0]
                lcp = int(neq[0]) if len(neq) else m
                neq = np.nonzero(R[::-1][:m] != Q[::-1][:m])[0]
                lcs = int(neq[0]) if len(neq) else m
                indel += lcp + lcs >= m
                other += lcp + lcs < m
    assert single >= need[0] and indel >= need[1] and other >= need[2], (single, indel, other)
    eng = Engine(params)
    try:
        got = eng.msa(arrs, n, nr, asm)
    finally:
        eng.close()
    bad = compare_vars(params, got, want, n)
    assert not bad, "\n".join(bad[:20])


from harness import compare_geno  # noqa: E402


@pytest.mark.parametrize("cfg,nwin,kw", [("C2", 6, {}), ("C3", 3, {}), ("C4", 1, dict(depths=(60, 60))),
                                          ("C5", 3, dict(num_samples=3)), ("C2", 3, dict(read_len=300)),
                                          ("C2", 2, dict(read_len=101, error_scale=4.0))])
def test_genotype_parity(cfg, nwin, kw):
    from lancet2_amd.engine import Engine
    ns = kw.pop("num_samples", 2)
    params = capi.default_params(min_k=25, max_k=25, num_samples=ns)
    arrs, n, nr = synth.make_config_batch(cfg, nwin, first_index=500, **kw)
    orc = OracleEngine(params)
    asm = orc.assemble(arrs, n, nr)
    var = orc.msa(arrs, n, nr, asm)
    want = orc.genotype(arrs, n, nr, asm, var)
    eng = Engine(params)
    try:
        got = eng.genotype(arrs, n, nr, asm, var)
    finally:
        eng.close()
    bad = compare_geno(params, got, want, n, nr, var["win_nvars"], arrs["read_win_off"])
    assert not bad, "\n".join(bad[:20])
    assert want["allele_counts"].sum() > 0 and (want["aln_rec"].reshape(-1, 6)[:, 0] > 0).sum() > 100


@pytest.mark.parametrize("cfg,nwin,kw", [("C2", 4, {}), ("C5", 2, dict(num_samples=3)), ("C2", 2, dict(snv_rate=1e-2, indel_rate=2e-3))])
def test_germline_quality_and_genotype_likelihoods(cfg, nwin, kw):
    """Outside case/control mode QUAL is the largest PL[0/0] of the samples with evidence (variant_call.cpp:289-303);
    PL / GQ are the Dirichlet-multinomial likelihoods over the allele depths (genotype_likelihood.cpp:93-272).
    Integers on both sides: compared exactly (north_star tolerance for QUAL: 1e-5)."""
    from lancet2_amd.engine import Engine
    ns = kw.pop("num_samples", 2)
    params = capi.default_params(min_k=25, max_k=25, num_samples=ns, case_ctrl_mode=0)
    arrs, n, nr = synth.make_config_batch(cfg, nwin, first_index=7300, **kw)
    orc = OracleEngine(params)
    asm = orc.assemble(arrs, n, nr)
    var = orc.msa(arrs, n, nr, asm)
    want = orc.genotype(arrs, n, nr, asm, var)
    eng = Engine(params)
    try:
        got = eng.genotype(arrs, n, nr, asm, var)
    finally:
        eng.close()
    bad = compare_geno(params, got, want, n, nr, var["win_nvars"], arrs["read_win_off"])
    assert not bad, "\n".join(bad[:20])
    assert np.abs(got["var_qual"] - want["var_qual"]).max() <= 1e-5
    assert want["var_qual"].max() > 20 and want["var_pl"].max() > 20 and want["var_gq"].max() > 0


@pytest.mark.parametrize("streams", [2, 3])
def test_process_batch_in_concurrent_lanes(streams):
    """ma_process_batch split into window ranges on separate streams gives the single-stream (= oracle) result."""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25)
    arrs, n, nr = synth.make_config_batch("C2", 7, first_index=900)
    orc = OracleEngine(params)
    want_a = orc.assemble(arrs, n, nr)
    want_v = orc.msa(arrs, n, nr, want_a)
    want_q = orc.genotype(arrs, n, nr, want_a, want_v)
    eng = Engine(params)
    try:
        eng.set_streams(streams)
        _, got_a, got_v, got_q = eng.process(arrs, n, nr, debug=True)
    finally:
        eng.close()
    bad = compare_asm(params, got_a, want_a, n) + compare_vars(params, got_v, want_v, n)
    bad += compare_geno(params, got_q, want_q, n, nr, want_v["win_nvars"], arrs["read_win_off"])
    assert not bad, "\n".join(bad[:20])


def test_process_batch_end_to_end():
    """gate -> assemble -> msa -> genotype chained on the device equals the oracle chain."""
    from lancet2_amd.engine import Engine
    params = capi.default_params()
    arrs, n, nr = synth.make_config_batch("C2", 5, first_index=700)
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
    assert not compare_asm(params, a, wa, n)
    assert not compare_vars(params, v, wv, n)
    bad = compare_geno(params, q, wq, n, nr, wv["win_nvars"], arrs["read_win_off"])
    assert not bad, "\n".join(bad[:20])


import glob  # noqa: E402
import os  # noqa: E402

_GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
                if not os.path.basename(p).startswith("pins_"))  # pins_*: brute-force score pins (test_aligner_pins.py)


@pytest.mark.parametrize("path", _GOLDEN, ids=[os.path.basename(p) for p in _GOLDEN])
def test_hip_path_matches_golden(path):
    """The committed fixtures (inputs + expected outputs of every stage) through ma_process_batch."""
    from lancet2_amd.engine import Engine
    from test_golden_and_shard import load_golden
    params, arrs, n, nr, want = load_golden(path)
    eng = Engine(params)
    try:
        g, a, v, q = eng.process(arrs, n, nr, debug=True)
        cx = eng.annotate(arrs, n, nr, a, v)
    finally:
        eng.close()
    assert np.array_equal(g["max_approx"], want["gate"]["max_approx"])
    assert not compare_asm(params, a, want["asm"], n)
    assert not compare_vars(params, v, want["var"], n)
    bad = compare_geno(params, q, want["geno"], n, nr, want["var"]["win_nvars"], arrs["read_win_off"])
    assert not bad, "\n".join(bad[:10])
    from harness import compare_cx
    compare_cx(params, cx, want["cx"], want["var"]["win_nvars"])


def test_full_size_properties():
    """BASELINE-sized batch (C2, 1024 windows): size-independent properties instead of an oracle run --
    tiling invariance (every replica of a window yields identical results), REF haplotype == reference
    anchor substring, read-support conservation (allele counts never exceed the reads of the sample)."""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25)
    base, n0, nr0 = synth.make_config_batch("C2", 16, first_index=4000)
    arrs, n, nr = synth.tile_batch(base, n0, nr0, 64)
    eng = Engine(params)
    try:
        g, a, v, q = eng.process(arrs, n, nr, debug=False)
    finally:
        eng.close()
    MH, ML, MC, MV = params.max_haps, params.max_hap_len, params.max_comps, params.max_vars
    S, NA = params.num_samples, params.max_alts + 1
    for name in ("win_status", "win_k", "win_ncomp"):
        x = a[name].reshape(64, n0)
        assert (x == x[0]).all(), name
    assert (v["win_nvars"].reshape(64, n0) == v["win_nvars"][:n0]).all()
    cnt = q["allele_counts"].reshape(64, -1)
    assert (cnt == cnt[0]).all()
    assert (a["win_ncomp"] > 0).mean() > 0.8
    reads_per_win = np.diff(arrs["read_win_off"].astype(np.int64))
    for w in range(n0):
        if a["win_ncomp"][w] == 0:
            continue
        ci = w * MC
        hi = w * MH + int(a["comp_hap0"][ci])
        L = int(a["hap_len"][hi])
        anchor = int(a["comp_anchor"][ci])
        ref = arrs["ref_bases"][int(arrs["ref_off"][w]): int(arrs["ref_off"][w + 1])]
        assert np.array_equal(a["hap_bases"][hi * ML: hi * ML + L], ref[anchor: anchor + L])
        per_var = q["allele_counts"][w * MV * S * NA * 2:(w + 1) * MV * S * NA * 2].reshape(MV, -1).sum(axis=1)
        assert per_var.max() <= reads_per_win[w]


def test_full_size_properties_c3_8192():
    """The bench's own shape -- C3 (60x/30x), 8192 windows per step, STR windows included, three concurrent lanes:
    tiling invariance across the 32 replicas of every window, REF haplotype == reference anchor substring,
    gated windows carry no result, and a checksum of the per-window checksums that is identical for every replica."""
    from lancet2_amd.engine import Engine
    import bench
    params = capi.default_params(min_k=25, max_k=25)
    base, n0, nr0 = bench.make_windows("C3", 256, 10_000, 8, 4)
    arrs, n, nr = synth.tile_batch(base, n0, nr0, 32)
    assert n == 8192
    eng = Engine(params)
    try:
        g, a, v, q = eng.process(arrs, n, nr, debug=False)
    finally:
        eng.close()
    MH, ML, MC, MV = params.max_haps, params.max_hap_len, params.max_comps, params.max_vars
    for name, d in (("win_status", a), ("win_k", a), ("win_ncomp", a), ("win_nvars", v), ("max_approx", g)):
        x = d[name].reshape(32, n0)
        assert (x == x[0]).all(), name
    sums = (q["allele_counts"].reshape(n, -1).astype(np.uint64) * np.arange(1, 1 + q["allele_counts"].size // n, dtype=np.uint64)).sum(axis=1)
    sums += a["hap_bases"].reshape(n, -1).astype(np.uint64).sum(axis=1) + v["allele_pool"].reshape(n, -1).astype(np.uint64).sum(axis=1)
    assert (sums.reshape(32, n0) == sums[:n0]).all()
    gated = g["max_approx"][:n0] >= 25
    assert gated.sum() >= 16 and (a["win_ncomp"][:n0][gated] == 0).all()
    assert ((a["win_status"][:n0] == 0).mean()) > 0.7  # (gated STR / duplication windows, tandem duplications at k = 25)
    for w in range(n0):
        if a["win_ncomp"][w] == 0:
            continue
        ci = w * MC
        hi = w * MH + int(a["comp_hap0"][ci])
        L = int(a["hap_len"][hi])
        anchor = int(a["comp_anchor"][ci])
        ref = arrs["ref_bases"][int(arrs["ref_off"][w]): int(arrs["ref_off"][w + 1])]
        assert np.array_equal(a["hap_bases"][hi * ML: hi * ML + L], ref[anchor: anchor + L])


def test_hints_never_change_results():
    """read_hint is a pure performance hint: absent, exact, shifted and random hints give identical
    assembly output (and all equal the oracle, which ignores hints)."""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25)
    arrs, n, nr = synth.make_config_batch("C2", 4, first_index=900)
    want = OracleEngine(params).assemble(arrs, n, nr)
    rng = np.random.default_rng(5)
    variants = {
        "exact": arrs["read_hint"].copy(),
        "none": None,
        "no_hint_value": np.full(nr, capi.MA_NO_HINT, dtype=np.int32),
        "shifted": (arrs["read_hint"] + rng.integers(-3, 4, nr)).astype(np.int32),
        "random": rng.integers(-400, 1400, nr).astype(np.int32),
        "huge": np.full(nr, 2_000_000_000, dtype=np.int32),
    }
    eng = Engine(params)
    try:
        for name, h in variants.items():
            a2 = dict(arrs)
            if h is None:
                a2.pop("read_hint")
            else:
                a2["read_hint"] = h
            got = eng.assemble(a2, n, nr)
            bad = compare_asm(params, got, want, n)
            assert not bad, name + ": " + "\n".join(bad[:10])
    finally:
        eng.close()


def test_mate_mer_dedup_corner_cases():
    """Overlapping mates, a read duplicated under the same qname, qnames shared across samples and a
    qname split into two non-adjacent runs: the hinted fast path must agree with the general path."""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25, num_samples=3)
    base = synth.make_window(950, W=1001, depths=(20, 20, 20), roles=(0, 0, 1))
    reads = base["reads"]
    passing = [r for r in reads if r["passf"]]
    # (1) fully overlapping mate: copy of a read with the same qname right after it
    extra = []
    for r in passing[:40:4]:
        c = dict(r)
        c["seq"] = r["seq"].copy()
        c["qual"] = r["qual"].copy()
        extra.append((r, c))
    # (2) same qname in two different samples of the same role (samples 0 and 1 are both normals)
    s0 = [r for r in passing if r["sample"] == 0][:10]
    s1 = [r for r in passing if r["sample"] == 1][:10]
    for x, y in zip(s0, s1):
        y["qname"] = x["qname"]
    out = []
    for r in reads:
        out.append(r)
        for orig, c in extra:
            if orig is r:
                out.append(c)
    # (3) a third copy far away from its group (non-adjacent run) -- keep collector order otherwise
    far = dict(passing[50])
    far["seq"] = passing[50]["seq"].copy()
    far["qual"] = passing[50]["qual"].copy()
    last_pass = max(i for i, r in enumerate(out) if r["passf"] and r["sample"] == passing[50]["sample"])
    out.insert(last_pass + 1, far)
    win = dict(ref=base["ref"], reads=out)
    arrs, n, nr = synth.pack_batch([win, synth.make_window(951, W=1001, depths=(20, 20, 20), roles=(0, 0, 1))])
    want = OracleEngine(params).assemble(arrs, n, nr)
    eng = Engine(params)
    try:
        got = eng.assemble(arrs, n, nr)
    finally:
        eng.close()
    bad = compare_asm(params, got, want, n)
    assert not bad, "\n".join(bad[:10])


# ---- SEQ_CX / GRAPH_CX annotation (core/variant_annotator.cpp:43-101) -----------------------------
from harness import compare_cx, handmade_annotation_case  # noqa: E402


@pytest.mark.parametrize("cfg,nwin,kw,gc", [("C2", 8, {}, 0.41), ("C2", 6, dict(str_unit=b"CA"), 0.41),
                                             ("C2", 6, dict(str_unit=b"A"), 0.5), ("C2", 4, dict(str_unit=b"CAG"), 0.41),
                                             ("C2", 3, dict(big_indel=60), 0.41), ("C5", 3, dict(num_samples=3), 0.3)])
def test_annotate_parity(cfg, nwin, kw, gc):
    from lancet2_amd.engine import Engine
    kw = dict(kw)
    ns = kw.pop("num_samples", 2)
    params = capi.default_params(min_k=25, max_k=37, num_samples=ns)
    arrs, n, nr = synth.make_config_batch(cfg, nwin, first_index=700, **kw)
    orc = OracleEngine(params)
    asm = orc.assemble(arrs, n, nr)
    var = orc.msa(arrs, n, nr, asm)
    want = orc.annotate(arrs, n, nr, asm, var, gc)
    eng = Engine(params)
    try:
        got = eng.annotate(arrs, n, nr, asm, var, gc)
    finally:
        eng.close()
    assert compare_cx(params, got, want, var["win_nvars"]) > 0


def _handmade_cases():
    rng = np.random.default_rng(77)

    def rnd(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, n))

    cases = []
    # the reference's own Score() fixtures (tests/base/sequence_complexity_test.cpp:152-222, :294-322)
    ref = "C" * 90 + "A" * 20 + "G" * 90
    cases.append(dict(haps=[ref, "C" * 90 + "A" * 25 + "G" * 85, "C" * 90 + "A" * 30 + "G" * 80], ref_pos=90, ref_len=20,
                      alts=[(25, {1: 90}), (30, {2: 90})]))
    acgt = "ACGT" * 50
    cases.append(dict(haps=[acgt, acgt], ref_pos=100, ref_len=1, alts=[(1, {1: 100})]))
    two = "ACGTACGTACGTACGTACGTACGTACGTACGT" + "TGCATGCATGCATGCATGCATGCATGCATGCA"
    cases.append(dict(haps=[two, two], ref_pos=16, ref_len=1, alts=[(1, {1: 16})]))
    cases.append(dict(haps=["A" * 200, "A" * 201], ref_pos=100, ref_len=1, alts=[(2, {1: 100})]))
    # no ALT site at all: Score(ref, ref) (variant_annotator.cpp:76-82)
    cases.append(dict(haps=[rnd(300), rnd(300)], ref_pos=150, ref_len=3, alts=[(1, {})]))
    # variant at the very start / end of short haplotypes, haplotype shorter than k = 7
    cases.append(dict(haps=["ACGTA", "ACTA"], ref_pos=0, ref_len=2, alts=[(1, {1: 0})]))
    r = rnd(120)
    cases.append(dict(haps=[r, r[:-1] + "A"], ref_pos=119, ref_len=1, alts=[(1, {1: 119})]))
    # N and lower-case bases (k-mer runs reset, entropy ignores them, repeats compare bytes)
    r = rnd(400)
    rn = r[:180] + "NNnn" + r[184:200] + "acacacacacacacac" + r[216:]
    cases.append(dict(haps=[rn, rn[:205] + rn[207:]], ref_pos=204, ref_len=3, alts=[(1, {1: 204})]))
    # stutter: 1-unit contraction of a dinucleotide repeat; imperfect trinucleotide repeat next to the site
    base = rnd(150)
    strr = base + "CA" * 14 + rnd(150)
    cases.append(dict(haps=[strr, base + "CA" * 13 + strr[178:]], ref_pos=149, ref_len=3, alts=[(1, {1: 149})]))
    tri = rnd(100) + "CAGCAACAGCAGCAGCTGCAG" + rnd(100)
    cases.append(dict(haps=[tri, tri[:98] + "T" + tri[99:]], ref_pos=98, ref_len=1, alts=[(1, {1: 98})]))
    # a 400-base deletion (window = 100 + 400 bases) and a 300-base insertion, several haplotypes per allele
    big = rnd(1500)
    dele = big[:500] + big[900:]
    ins = big[:700] + rnd(300) + big[700:]
    cases.append(dict(haps=[big, dele, dele, ins], ref_pos=499, ref_len=401, alts=[(1, {1: 499, 2: 499})]))
    cases.append(dict(haps=[big, dele, dele, ins], ref_pos=699, ref_len=1, alts=[(301, {3: 699})]))
    # multi-allelic with hexamer and homopolymer contexts
    hexa = rnd(200) + "TTAGGG" * 9 + rnd(200)
    cases.append(dict(haps=[hexa, hexa[:254] + hexa[260:], hexa[:230] + "G" + hexa[231:], hexa[:254] + "TTAGGG" + hexa[254:]],
                      ref_pos=229, ref_len=31, alts=[(25, {1: 229}), (31, {2: 229}), (37, {3: 229})],
                      cx=(60, 55, 4), cxf=(0.01, 1.75, 0.5)))
    return cases


@pytest.mark.parametrize("gc", [0.41, 0.5])
def test_annotate_handmade_cases(gc):
    """Edge cases the synthetic windows never produce, driven straight through ma_annotate_batch."""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25)
    cases = _handmade_cases()
    asm, var = handmade_annotation_case(params, cases)
    n = len(cases)
    arrs = dict(ref_off=np.zeros(n + 1, np.uint32), read_win_off=np.zeros(n + 1, np.uint32))
    orc = OracleEngine(params)
    want = orc.annotate(arrs, n, 0, asm, var, gc)
    eng = Engine(params)
    try:
        got = eng.annotate(arrs, n, 0, asm, var, gc)
    finally:
        eng.close()
    assert compare_cx(params, got, want, var["win_nvars"]) == n
    # sanity against the reference's expectations for its own fixtures
    assert want["seq_cx_i"][0] >= 20 and want["seq_cx_i"].reshape(n, -1, 4)[1, 0, 0] == 1


@pytest.mark.parametrize("cfg,nwin,first,kw", [("C2", 40, 20_000, {}), ("C3", 10, 30_000, {}),
                                                ("C2", 12, 40_000, dict(error_scale=3.0, indel_rate=1e-3)),
                                                ("C2", 8, 50_000, dict(str_unit=b"CAG", n_somatic=3))])
def test_process_batch_seed_sweep(cfg, nwin, first, kw):
    """Many more seeds than the targeted cases above: the whole chain (+ annotation) against the oracle, with the
    default settings (read hints, banded POA kernel, automatic lanes)."""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=37)
    arrs, n, nr = synth.make_config_batch(cfg, nwin, first_index=first, **kw)
    orc = OracleEngine(params)
    wg = orc.gate(arrs, n, nr)
    wa = orc.assemble(arrs, n, nr)
    wv = orc.msa(arrs, n, nr, wa)
    wq = orc.genotype(arrs, n, nr, wa, wv)
    wc = orc.annotate(arrs, n, nr, wa, wv)
    eng = Engine(params)
    try:
        g, a, v, q = eng.process(arrs, n, nr, debug=True)
        cx = eng.annotate(arrs, n, nr, a, v)
    finally:
        eng.close()
    assert np.array_equal(g["max_approx"], wg["max_approx"]) and np.array_equal(g["max_exact"], wg["max_exact"])
    bad = compare_asm(params, a, wa, n) + compare_vars(params, v, wv, n)
    bad += compare_geno(params, q, wq, n, nr, wv["win_nvars"], arrs["read_win_off"])
    assert not bad, "\n".join(bad[:20])
    compare_cx(params, cx, wc, wv["win_nvars"])


@pytest.mark.parametrize("depth", [300, 500])
def test_process_batch_deep_window(depth):
    """A 300x/300x and a 500x/500x (BASELINE configs[3]) panel-style window with 50 bp indels (4288 / 7100 reads): more
    sequences than the LDS-resident shortcuts of the build stage accept (k_mm_lds falls back to the HBM mate-mer set),
    four haplotypes, 13 variants, read<->haplotype regions widened by the 50-base indels."""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25)
    arrs, n, nr = synth.make_config_batch("C4", 1, first_index=60000, depths=(depth, depth))
    assert nr > 2048
    orc = OracleEngine(params)
    wa = orc.assemble(arrs, n, nr)
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
    assert int(wv["win_nvars"][0]) >= 8


@pytest.mark.parametrize("tier,kw", [(1, {}), (2, dict(big_indel=50)), (3, dict(big_indel=80, depths=(40, 40))),
                                     (0, dict(big_indel=80, depths=(40, 40), read_len=250))])
def test_genotype_parity_alignment_tiers(tier, kw):
    """aln_tier is a testing knob that results must not depend on: bit 0 sends every DP pair through the any-width
    kernel (row in LDS), bit 1 switches the gapless certificates off so that every seeded pair runs the DP.  Long
    indels (BASELINE configs[3]) and 250-base reads widen the search regions into the upper width classes."""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25, aln_tier=tier)
    arrs, n, nr = synth.make_config_batch("C2", 3, first_index=91_000, **kw)
    orc = OracleEngine(params)
    asm = orc.assemble(arrs, n, nr)
    var = orc.msa(arrs, n, nr, asm)
    want = orc.genotype(arrs, n, nr, asm, var)
    eng = Engine(params)
    try:
        got = eng.genotype(arrs, n, nr, asm, var)
    finally:
        eng.close()
    bad = compare_geno(params, got, want, n, nr, var["win_nvars"], arrs["read_win_off"])
    assert not bad, "\n".join(bad[:20])
    assert (want["aln_rec"].reshape(-1, 6)[:, 0] > 0).sum() > 100


@pytest.mark.parametrize("streams,taps", [(1, False), (1, True), (2, True)])
def test_device_memspace_matches_host_memspace(streams, taps):
    """MA_MEM_DEVICE (caller-owned device buffers, what bench.py times) gives the bytes of MA_MEM_HOST."""
    from harness import DeviceArena
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25)
    arrs, n, nr = synth.make_config_batch("C2", 6, first_index=95_000)
    eng = Engine(params)
    try:
        hg, ha, hv, hq = eng.process(arrs, n, nr, debug=taps)
    finally:
        eng.close()
    specs = [capi.gate_out_spec(n), capi.asm_out_spec(params, n), capi.var_out_spec(params, n),
             capi.geno_out_spec(params, n, nr, debug=taps)]
    deng = Engine(params, memspace=capi.MA_MEM_DEVICE)
    arena = DeviceArena()
    try:
        b = capi.make_batch_struct({k: arena.upload(v) for k, v in arrs.items()}, n, nr)
        ptrs = [{k: arena.alloc(int(sz) * np.dtype(dt).itemsize) for k, (dt, sz) in spec.items()} for spec in specs]
        deng.set_streams(streams)
        deng.process_device(b, capi.fill_struct(capi.GateOut, ptrs[0]), capi.fill_struct(capi.AsmOut, ptrs[1]),
                            capi.fill_struct(capi.VarOut, ptrs[2]), capi.fill_struct(capi.GenoOut, ptrs[3]))
        deng.synchronize()
        dg, da, dv, dq = [{k: arena.download(pt[k], dt, sz) for k, (dt, sz) in spec.items()}
                          for spec, pt in zip(specs, ptrs)]
    finally:
        deng.close()
        arena.close()
    assert np.array_equal(dg["max_approx"], hg["max_approx"])
    # the helpers compare the defined part of every buffer (unused slots of a capacity-sized array are not written)
    bad = compare_asm(params, da, ha, n) + compare_vars(params, dv, hv, n)
    if taps:
        bad += compare_geno(params, dq, hq, n, nr, hv["win_nvars"], arrs["read_win_off"])
    assert not bad, "\n".join(bad[:10])
    per = params.num_samples * (params.max_alts + 1) * 2
    diff = np.nonzero(dq["allele_counts"] != hq["allele_counts"])[0]
    assert diff.size == 0, "allele_counts differ at (window, variant, slot) " + str(
        [(int(x) // (per * params.max_vars), int(x) // per % params.max_vars, int(x) % per, int(dq["allele_counts"][x]),
          int(hq["allele_counts"][x])) for x in diff[:12]]) + " win_nvars " + str(hv["win_nvars"].tolist())
    assert np.array_equal(dq["var_qual"].view(np.uint64), hq["var_qual"].view(np.uint64))


@pytest.mark.parametrize("kw,pk", [(dict(tandem_dup=40), {}), (dict(tandem_dup=60), dict(min_k=25, max_k=25)),
                                   (dict(dup_len=260), dict(min_k=25, max_k=25)),
                                   (dict(low_complexity=100, softclip_frac=0.08, n_frac=0.04), {})])
def test_whole_chain_parity_on_the_harder_bench_shapes(kw, pk):
    """Tandem duplications in the sample (a repeat longer than k: cycles until the k ladder outgrows it), dispersed
    duplications of the reference (the gate skips every k), low-complexity stretches,
    soft-clipped and N-containing reads -- the shapes bench.py mixes into its WGS windows since round 3: every stage
    bit-identical to the oracle, with the reference's k cascade and at the bench's single k = 25."""
    from lancet2_amd.engine import Engine
    params = capi.default_params(**pk)
    arrs, n, nr = synth.make_config_batch("C3", 4, first_index=83_000, **kw)
    orc = OracleEngine(params)
    wa = orc.assemble(arrs, n, nr)
    wv = orc.msa(arrs, n, nr, wa)
    wq = orc.genotype(arrs, n, nr, wa, wv)
    eng = Engine(params)
    try:
        g, a, v, q = eng.process(arrs, n, nr, debug=True)
    finally:
        eng.close()
    bad = compare_asm(params, a, wa, n) + compare_vars(params, v, wv, n)
    bad += compare_geno(params, q, wq, n, nr, wv["win_nvars"], arrs["read_win_off"])
    assert not bad, "\n".join(bad[:12])
    if "tandem_dup" in kw and not pk:  # the duplication sent at least one window up the ladder
        assert (wa["win_k"] > 40).any(), wa["win_k"].tolist()


def test_mate_mer_set_that_fills_up_is_split_not_truncated(monkeypatch):
    """k_mm_lds keeps a window's (k-mer, read pair) keys in an LDS set, one class of table slots per pass.  The classes are
    sized by an instance COUNT; the keys of a k-mer that many read pairs carry all fall into one class, so a class can
    outgrow the set.  A window whose set fills up is flagged like any other capacity and the retry pass sends its mate-mers
    through the HBM set -- forced here by declaring the set full after six probes: same graphs, same supports, same calls as
    the oracle, no flag left."""
    from lancet2_amd.engine import Engine
    monkeypatch.setenv("MA_MM_PROBE_MAX", "6")
    params = capi.default_params(min_k=25, max_k=25)
    arrs, n, nr = synth.make_config_batch("C3", 8, first_index=87_500, softclip_frac=0.05, n_frac=0.03)
    orc = OracleEngine(params)
    wa = orc.assemble(arrs, n, nr)
    wv = orc.msa(arrs, n, nr, wa)
    wq = orc.genotype(arrs, n, nr, wa, wv)
    eng = Engine(params)
    try:
        g, a, v, q = eng.process(arrs, n, nr, debug=True)
    finally:
        eng.close()
    assert not (a["win_status"] & capi.MA_W_TABLE_OVERFLOW).any()
    bad = compare_asm(params, a, wa, n) + compare_vars(params, v, wv, n)
    bad += compare_geno(params, q, wq, n, nr, wv["win_nvars"], arrs["read_win_off"])
    assert not bad, "\n".join(bad[:12])


@pytest.mark.parametrize("pk", ["1", "0"])
def test_packed_two_pairs_per_lane_aligner_is_the_one_pair_aligner(pk, monkeypatch):
    """MA_ALIGN_PK=1: the two busiest register classes of the read aligner run two pairs per lane on packed 16-bit halves
    (align.hip: k_align_reg2p -- decisions as bit planes, walks back in k_align_tb2) for the pairs whose region cannot reach a
    haplotype end, the general body beside them for the others.  Same records as the oracle, on WGS-shaped windows with
    tandem repeats, soft clips and N bases; the timing names prove the route was taken."""
    from lancet2_amd.engine import Engine
    monkeypatch.setenv("MA_ALIGN_PK", pk)  # (round 5: the packed launch is the default; "0" keeps the one-pair launch covered)
    monkeypatch.setenv("MA_NO_REROUTE", "1")  # (a class of a few hundred pairs would go to the wavefront kernel: no two classes to pair)
    params = capi.default_params(min_k=25, max_k=25)
    arrs, n, nr = synth.make_config_batch("C3", 12, first_index=86_000, softclip_frac=0.05, n_frac=0.03)
    orc = OracleEngine(params)
    wa = orc.assemble(arrs, n, nr)
    wv = orc.msa(arrs, n, nr, wa)
    wq = orc.genotype(arrs, n, nr, wa, wv)
    eng = Engine(params)
    try:
        eng.timing_control(1)
        g, a, v, q = eng.process(arrs, n, nr, debug=True)
        names = {k for k, _ in eng.kernel_times()}
    finally:
        eng.close()
    assert ("k_align_tb" in names) == (pk == "1"), sorted(names)
    bad = compare_asm(params, a, wa, n) + compare_vars(params, v, wv, n)
    bad += compare_geno(params, q, wq, n, nr, wv["win_nvars"], arrs["read_win_off"])
    assert not bad, "\n".join(bad[:12])


def test_ladder_tail_six_rungs_at_a_time_equals_rung_by_rung(monkeypatch):
    """Once few windows are left on the k ladder their next six rungs are attempted at once (assemble.hip: speculate_tail --
    the pending windows copied into a derived batch, one copy per rung, each window taking the first rung that resolved).
    Same bytes and the same attempt count as the pass-per-rung loop (MA_NO_SPEC) and as the oracle: tandem duplications of
    30 to 110 bases (ladders of 4 to 17 rungs, two rounds of six and more), a window that climbs to the top and stays
    unresolved, windows that resolve on the first rung, and a ladder with a coarser step."""
    from lancet2_amd.engine import Engine
    wins = []
    for i, dup in enumerate((30, 45, 64, 80, 110, 0, 0, 95)):
        kw = dict(synth.CONFIGS["C2"])
        if dup:
            kw["tandem_dup"] = dup
        wins.append(synth.make_window(84_000 + i, **kw))
    arrs, n, nr = synth.pack_batch(wins)
    for pk in (dict(), dict(min_k=13, max_k=61, k_step=12)):
        params = capi.default_params(**pk)
        orc = OracleEngine(params)
        want = orc.assemble(arrs, n, nr)
        got = {}
        for mode in ("spec", "rungs"):
            if mode == "rungs":
                monkeypatch.setenv("MA_NO_SPEC", "1")
            else:
                monkeypatch.delenv("MA_NO_SPEC", raising=False)
            eng = Engine(params)
            try:
                eng.timing_control(1)
                a = eng.assemble(arrs, n, nr)
                got[mode] = (a, eng.stats()["window_attempts"], sum(1 for k_, _ in eng.kernel_times() if k_ == "k_clean"))
            finally:
                eng.close()
        monkeypatch.delenv("MA_NO_SPEC", raising=False)
        for mode in got:
            bad = compare_asm(params, got[mode][0], want, n)
            assert not bad, (pk, mode, bad[:8])
        assert got["spec"][1] == got["rungs"][1], (pk, got["spec"][1], got["rungs"][1])
        assert got["spec"][2] < got["rungs"][2], (pk, got["spec"][2], got["rungs"][2])   # fewer passes is the point
        assert (want["win_k"] > 50).any() and (want["win_k"] <= 31).any(), want["win_k"].tolist()


def test_truncated_cigars_are_flagged_not_silent():
    """The alignment records hold max_cigar operations and the scoring epilogue (local_scorer.cpp:166-279) walks the whole
    CIGAR: with a cap that some read's CIGAR exceeds (max_cigar = 4 on indel-dense windows) every window that holds such
    a read says so (MA_W_CIGAR_OVERFLOW) and every other window equals the oracle; with a cap that suffices (64) no window
    is flagged and all of them equal the oracle."""
    from lancet2_amd.engine import Engine
    arrs, n, nr = synth.make_config_batch("C2", 4, first_index=8800, indel_rate=1.5e-2, snv_rate=2e-3)
    rwo = arrs["read_win_off"]
    for mcg in (4, 64):
        params = capi.default_params(min_k=25, max_k=25, max_cigar=mcg)
        orc = OracleEngine(params)
        asm = orc.assemble(arrs, n, nr)
        var = orc.msa(arrs, n, nr, asm)
        want = orc.genotype(arrs, n, nr, asm, var)
        status0 = asm["win_status"].copy()
        eng = Engine(params)
        try:
            got = eng.genotype(arrs, n, nr, asm, var)
        finally:
            eng.close()
        ncig = want["aln_cigar"].reshape(nr, params.max_haps, 1 + mcg)[:, :, 0]
        long_win = np.array([bool((ncig[rwo[w]:rwo[w + 1]] > mcg).any()) for w in range(n)])
        flagged = (asm["win_status"] & capi.MA_W_CIGAR_OVERFLOW) != 0
        assert np.array_equal(asm["win_status"] & ~np.uint32(capi.MA_W_CIGAR_OVERFLOW), status0)
        # a flag needs a long CIGAR of a read that overlaps a variant; no long CIGAR, no flag
        assert not (flagged & ~long_win).any()
        if mcg == 4:
            assert flagged.any(), "the batch was meant to hold CIGARs of more than four operations"
        else:
            assert not flagged.any()
        per = params.max_vars * params.num_samples * (params.max_alts + 1) * 2
        gc, wc = got["allele_counts"].reshape(n, per), want["allele_counts"].reshape(n, per)
        for w in range(n):
            if not flagged[w]:
                assert np.array_equal(gc[w], wc[w]), f"window {w} (max_cigar {mcg})"
        assert want["allele_counts"].sum() > 0


def test_over_long_reads_flag_their_window_not_the_batch():
    """A read beyond what a stage supports (genotyping: 608 bases, assembly: 1024) used to fail the whole batch with
    MA_ERR_PARAM; now its window is skipped by that stage and says so (MA_W_READ_OVERFLOW), the other windows equal the
    oracle."""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25)
    kw = dict(synth.CONFIGS["C2"])
    wins = [synth.make_window(9100 + i, **kw) for i in range(4)]
    wins[1] = synth.make_window(9101, **dict(kw, read_len=700, depths=(12, 12)))
    wins[3] = synth.make_window(9103, **dict(kw, read_len=1100, depths=(8, 8)))
    arrs, n, nr = synth.pack_batch(wins)
    eng = Engine(params)
    try:
        g, a, v, q = eng.process(arrs, n, nr, debug=False)
    finally:
        eng.close()
    orc = OracleEngine(params)
    wa = orc.assemble(arrs, n, nr)
    wv = orc.msa(arrs, n, nr, wa)
    wq = orc.genotype(arrs, n, nr, wa, wv, debug=False)
    st = a["win_status"]
    assert st[3] & capi.MA_W_READ_OVERFLOW and st[3] & capi.MA_W_NO_HAPLOTYPE      # not assembled
    assert st[1] & capi.MA_W_READ_OVERFLOW and not (st[1] & capi.MA_W_NO_HAPLOTYPE)  # assembled, not genotyped
    assert not (st[0] & capi.MA_W_READ_OVERFLOW) and not (st[2] & capi.MA_W_READ_OVERFLOW)
    per = params.max_vars * params.num_samples * (params.max_alts + 1) * 2
    gc, wc = q["allele_counts"].reshape(n, per), wq["allele_counts"].reshape(n, per)
    for w in (0, 2):
        assert np.array_equal(gc[w], wc[w]) and wc[w].sum() > 0
    assert gc[1].sum() == 0 and gc[3].sum() == 0
    # window 1's haplotypes and variants are the oracle's
    from harness import compare_vars as cv
    sel = np.array([0, 1, 2])
    assert int(v["win_nvars"][1]) == int(wv["win_nvars"][1]) and int(a["win_ncomp"][1]) == int(wa["win_ncomp"][1])


@pytest.mark.parametrize("streams", [1, 3])
def test_host_route_packed_results_equal_the_whole_array_route(streams, monkeypatch):
    """MA_MEM_HOST: every lane uploads its own slice and brings back packed records of what it wrote (pack.hip), scattered
    into the caller's fixed-stride arrays on the host -- against the old route that copied every array whole
    (MA_HOST_LEGACY): the same bytes in every array, including the taps, on a batch whose windows differ in everything
    (no reads, sub-anchor coverage, several components, 60-base indels)."""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25)
    kw = dict(synth.CONFIGS["C2"])
    wins = [synth.make_window(9300 + i, **kw) for i in range(9)]
    wins[2] = synth.make_window(9302, **dict(kw, depths=(2, 2)))          # below the anchor coverage
    wins[4] = synth.make_window(9304, **dict(kw, big_indel=60))
    wins[5] = dict(ref=wins[5]["ref"], reads=[])                            # no reads at all
    wins[7] = synth.make_window(9307, **dict(kw, snv_rate=8e-3, indel_rate=2e-3))
    arrs, n, nr = synth.pack_batch(wins)
    monkeypatch.setenv("MA_HOST_LEGACY", "1")
    eng = Engine(params)
    try:
        want = eng.process(arrs, n, nr, debug=True)
    finally:
        eng.close()
    monkeypatch.delenv("MA_HOST_LEGACY")
    eng = Engine(params)
    try:
        eng.set_streams(streams)
        got = eng.process(arrs, n, nr, debug=True)
    finally:
        eng.close()
    g, a, v, q = got
    wg, wa, wv, wq = want
    assert np.array_equal(g["max_approx"], wg["max_approx"]) and np.array_equal(g["max_exact"], wg["max_exact"])
    bad = compare_asm(params, a, wa, n) + compare_vars(params, v, wv, n)
    bad += compare_geno(params, q, wq, n, nr, wv["win_nvars"], arrs["read_win_off"])
    assert not bad, "\n".join(bad[:12])
    assert wv["win_nvars"].sum() > 5 and (wa["win_ncomp"] == 0).any()


def test_prefetched_batches_give_the_same_results():
    """ma_prefetch_batch uploads the next batch while this one computes (double-buffered input sets, one copy stream per
    context).  Results must not depend on it: three different batches processed in turn, each prefetched before the
    previous one is processed; a batch processed without having been prefetched in between; a third prefetch while two
    are waiting (ignored); the same batch prefetched and processed twice."""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25)
    batches = [synth.make_config_batch("C2", 5, first_index=9400), synth.make_config_batch("C3", 3, first_index=9500),
               synth.make_config_batch("C2", 7, first_index=9600, indel_rate=1e-3)]
    eng = Engine(params)
    try:
        want = [eng.process(a_, n_, nr_, debug=False) for a_, n_, nr_ in batches]
    finally:
        eng.close()

    def run(eng, i):
        arrs, n, nr = batches[i]
        outs = (capi.alloc_host(capi.gate_out_spec(n)), capi.alloc_host(capi.asm_out_spec(params, n)),
                capi.alloc_host(capi.var_out_spec(params, n)), capi.alloc_host(capi.geno_out_spec(params, n, nr, False)))
        eng.process_device(structs[i], capi.fill_struct(capi.GateOut, outs[0]), capi.fill_struct(capi.AsmOut, outs[1]),
                           capi.fill_struct(capi.VarOut, outs[2]), capi.fill_struct(capi.GenoOut, outs[3]))
        g, a, v, q = outs
        wg, wa, wv, wq = want[i]
        assert np.array_equal(g["max_approx"], wg["max_approx"])
        bad = compare_asm(params, a, wa, n) + compare_vars(params, v, wv, n)
        assert not bad, (i, bad[:6])
        assert np.array_equal(q["allele_counts"], wq["allele_counts"]) and np.array_equal(
            q["var_qual"].view(np.uint64), wq["var_qual"].view(np.uint64))

    structs = [capi.make_batch_struct(a_, n_, nr_) for a_, n_, nr_ in batches]
    eng = Engine(params)
    try:
        eng.prefetch(structs[0])
        eng.prefetch(structs[1])
        eng.prefetch(structs[2])      # two are waiting: ignored
        run(eng, 0)
        eng.prefetch(structs[2])
        run(eng, 1)
        run(eng, 2)
        run(eng, 1)                   # never prefetched this time
        eng.prefetch(structs[0])
        eng.prefetch(structs[0])
        run(eng, 0)
        run(eng, 0)
        eng.prefetch(structs[2])      # prefetched, never processed: the engine must close cleanly
    finally:
        eng.close()


def test_jobs_queued_ahead_follow_the_call_that_collects_them():
    """The host route queues a prefetched batch's compute jobs behind the running ones (the lanes are persistent worker
    threads), assuming the call that brings the batch asks for what the last call asked for.  When it does not -- other
    optional outputs, the per-read debug taps, kernel timing switched in between -- the queued results are dropped and the
    batch is computed in the call.  Outputs the caller leaves out stay untouched; counters and kernel timers are per call."""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25)
    batches = [synth.make_config_batch("C2", 6, first_index=9700), synth.make_config_batch("C3", 4, first_index=9800)]
    eng = Engine(params)
    try:
        want = [eng.process(a_, n_, nr_, debug=True) for a_, n_, nr_ in batches]
    finally:
        eng.close()
    structs = [capi.make_batch_struct(a_, n_, nr_) for a_, n_, nr_ in batches]

    def run(eng, i, leave_out=(), debug=False):
        arrs, n, nr = batches[i]
        outs = [capi.alloc_host(capi.gate_out_spec(n)), capi.alloc_host(capi.asm_out_spec(params, n)),
                capi.alloc_host(capi.var_out_spec(params, n)), capi.alloc_host(capi.geno_out_spec(params, n, nr, debug))]
        for d_ in outs:
            for k_ in leave_out:
                if k_ in d_:
                    d_[k_].view(np.uint8)[...] = 0x5A
        kept = [{k_: v_ for k_, v_ in d_.items() if k_ not in leave_out} for d_ in outs]
        eng.process_device(structs[i], capi.fill_struct(capi.GateOut, kept[0]), capi.fill_struct(capi.AsmOut, kept[1]),
                           capi.fill_struct(capi.VarOut, kept[2]), capi.fill_struct(capi.GenoOut, kept[3]))
        g, a, v, q = outs
        wg, wa, wv, wq = want[i]
        assert np.array_equal(g["max_approx"], wg["max_approx"]) and np.array_equal(g["max_exact"], wg["max_exact"])
        for k_ in leave_out:  # nothing was written where the caller passed no array
            for d_ in outs:
                if k_ in d_:
                    assert (d_[k_].view(np.uint8) == 0x5A).all(), k_
        if not leave_out:
            bad = compare_asm(params, a, wa, n) + compare_vars(params, v, wv, n)
            if debug:
                bad += compare_geno(params, q, wq, n, nr, wv["win_nvars"], arrs["read_win_off"])
            assert not bad, (i, bad[:6])
        else:
            assert np.array_equal(a["win_status"], wa["win_status"]) and np.array_equal(v["win_nvars"], wv["win_nvars"])
        assert np.array_equal(q["allele_counts"], wq["allele_counts"])
        assert np.array_equal(q["var_qual"].view(np.uint64), wq["var_qual"].view(np.uint64))

    eng = Engine(params)
    try:
        eng.timing_control(1)
        run(eng, 0)                                   # the first call says what callers ask for
        t0 = dict(eng.kernel_times())
        assert t0.get("k_clean", 0) > 0 and t0.get("k_vote", 0) > 0
        eng.prefetch(structs[1])                      # queued ahead, full outputs
        run(eng, 1, leave_out=("var_pl", "var_gq", "hap_runs", "comp_cxf", "alt_type"))   # ... but the call wants fewer
        eng.prefetch(structs[0])                      # queued ahead with the reduced set
        run(eng, 0)                                   # ... the call wants everything again
        eng.prefetch(structs[1])
        run(eng, 1, debug=True)                       # the per-read taps are never computed ahead
        eng.prefetch(structs[0])
        eng.timing_control(0)                         # switched while a job is queued: waits for it, the job is then dropped
        run(eng, 0)
        assert eng.kernel_times() == []
        eng.prefetch(structs[1])
        eng.prefetch(structs[0])
        run(eng, 1)
        run(eng, 0)                                   # two jobs queued ahead, collected in order
        eng.timing_control(1)
        eng.prefetch(structs[1])
        run(eng, 1)
        t1 = dict(eng.kernel_times())                 # the timers of THIS call's batch, not of whatever runs next
        assert t1.get("k_clean", 0) > 0
        assert eng.stats()["pairs"] > 0
    finally:
        eng.close()


def test_queued_ahead_jobs_survive_a_change_of_lanes_and_tiny_batches():
    """The pipelined host route with batches of one and two windows (one lane), an empty window range in a lane, and the
    number of lanes changed between ma_prefetch_batch and the call that brings the batch (the queued job was split for the
    old number: it is left alone and the batch is computed for the new one)."""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25)
    batches = [synth.make_config_batch("C2", 1, first_index=9900), synth.make_config_batch("C2", 2, first_index=9910),
               synth.make_config_batch("C2", 9, first_index=9920)]
    eng = Engine(params)
    try:
        want = [eng.process(a_, n_, nr_, debug=False) for a_, n_, nr_ in batches]
    finally:
        eng.close()
    structs = [capi.make_batch_struct(a_, n_, nr_) for a_, n_, nr_ in batches]

    def run(eng, i):
        arrs, n, nr = batches[i]
        outs = (capi.alloc_host(capi.gate_out_spec(n)), capi.alloc_host(capi.asm_out_spec(params, n)),
                capi.alloc_host(capi.var_out_spec(params, n)), capi.alloc_host(capi.geno_out_spec(params, n, nr, False)))
        eng.process_device(structs[i], capi.fill_struct(capi.GateOut, outs[0]), capi.fill_struct(capi.AsmOut, outs[1]),
                           capi.fill_struct(capi.VarOut, outs[2]), capi.fill_struct(capi.GenoOut, outs[3]))
        g, a, v, q = outs
        wg, wa, wv, wq = want[i]
        assert np.array_equal(g["max_approx"], wg["max_approx"])
        bad = compare_asm(params, a, wa, n) + compare_vars(params, v, wv, n)
        assert not bad, (i, bad[:6])
        assert np.array_equal(q["allele_counts"], wq["allele_counts"])

    eng = Engine(params)
    try:
        run(eng, 0)
        eng.prefetch(structs[1])
        run(eng, 1)
        eng.prefetch(structs[0])
        eng.prefetch(structs[2])
        run(eng, 0)
        eng.set_streams(3)            # the job queued for batch 2 was split for one lane
        run(eng, 2)
        eng.prefetch(structs[2])      # queued for three lanes ...
        eng.set_streams(4)            # ... four are asked for: nine windows, lanes of 2-3
        run(eng, 2)
        eng.set_streams(8)
        eng.prefetch(structs[1])      # two windows: one lane whatever was asked for
        run(eng, 1)
        eng.prefetch(structs[2])
        run(eng, 2)
    finally:
        eng.close()


@pytest.mark.parametrize("seed", [20261003, 7, 4242])
def test_host_route_pipeline_under_a_random_sequence_of_calls(seed):
    """A seeded random walk over what a caller can do with one MA_MEM_HOST context -- prefetch a batch (or the same one
    twice, or a third while two wait), process a batch that was or was not prefetched, leave optional outputs out, switch
    the timing mode, change the number of lanes, read counters and timers in between -- against results computed once up
    front: the lanes' worker threads, the uploader thread and the calling thread must agree whatever the order."""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25)
    batches = [synth.make_config_batch("C2", 3, first_index=9950), synth.make_config_batch("C3", 5, first_index=9960),
               synth.make_config_batch("C2", 8, first_index=9970, indel_rate=1e-3), synth.make_config_batch("C2", 1, first_index=9980)]
    eng = Engine(params)
    try:
        want = [eng.process(a_, n_, nr_, debug=False) for a_, n_, nr_ in batches]
    finally:
        eng.close()
    structs = [capi.make_batch_struct(a_, n_, nr_) for a_, n_, nr_ in batches]
    optional = ("var_pl", "var_gq", "hap_runs", "comp_cx", "alt_length", "hap_stats")
    rng = np.random.default_rng(seed)

    def run(eng, i, leave_out):
        arrs, n, nr = batches[i]
        outs = [capi.alloc_host(capi.gate_out_spec(n)), capi.alloc_host(capi.asm_out_spec(params, n)),
                capi.alloc_host(capi.var_out_spec(params, n)), capi.alloc_host(capi.geno_out_spec(params, n, nr, False))]
        kept = [{k_: v_ for k_, v_ in d_.items() if k_ not in leave_out} for d_ in outs]
        eng.process_device(structs[i], capi.fill_struct(capi.GateOut, kept[0]), capi.fill_struct(capi.AsmOut, kept[1]),
                           capi.fill_struct(capi.VarOut, kept[2]), capi.fill_struct(capi.GenoOut, kept[3]))
        g, a, v, q = outs
        wg, wa, wv, wq = want[i]
        assert np.array_equal(g["max_approx"], wg["max_approx"]), i
        assert np.array_equal(a["win_status"], wa["win_status"]) and np.array_equal(v["win_nvars"], wv["win_nvars"]), i
        assert np.array_equal(a["hap_len"], wa["hap_len"]) and np.array_equal(q["allele_counts"], wq["allele_counts"]), i
        if not leave_out:
            bad = compare_asm(params, a, wa, n) + compare_vars(params, v, wv, n)
            assert not bad, (i, bad[:6])

    eng = Engine(params)
    try:
        for step in range(60):
            op = int(rng.integers(0, 10))
            i = int(rng.integers(0, len(batches)))
            if op <= 3:
                eng.prefetch(structs[i])
            elif op <= 7:
                lo = tuple(k_ for k_ in optional if rng.random() < 0.25) if rng.random() < 0.4 else ()
                run(eng, i, lo)
            elif op == 8:
                eng.timing_control(int(rng.integers(0, 3)))
                eng.kernel_times()
                eng.stats()
            else:
                eng.set_streams(int(rng.integers(0, 5)))
        run(eng, 2, ())
    finally:
        eng.close()


def test_device_buffers_need_no_padding_or_alignment():
    """MA_MEM_DEVICE passes the caller's pointers through: the byte arrays (reference, read bases, qualities) sized
    exactly -- without the 64 bytes of padding the host route adds --, at odd addresses, with junk on both sides, give the
    same bytes (k_insert stages the window's reads with aligned 8-byte words and must not let what lies outside in)."""
    from harness import DeviceArena
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25)
    arrs, n, nr = synth.make_config_batch("C2", 5, first_index=95_500)
    eng = Engine(params)
    try:
        hg, ha, hv, hq = eng.process(arrs, n, nr, debug=False)
    finally:
        eng.close()
    specs = [capi.gate_out_spec(n), capi.asm_out_spec(params, n), capi.var_out_spec(params, n),
             capi.geno_out_spec(params, n, nr, debug=False)]
    deng = Engine(params, memspace=capi.MA_MEM_DEVICE)
    arena = DeviceArena()
    try:
        dptr = {}
        for k, v in arrs.items():
            if k in ("ref_bases", "read_bases", "read_quals"):
                dptr[k] = arena.upload_unaligned(v[:-64], shift={"ref_bases": 1, "read_bases": 3, "read_quals": 5}[k])
            else:
                dptr[k] = arena.upload(v)
        b = capi.make_batch_struct(dptr, n, nr)
        ptrs = [{k: arena.alloc(int(sz) * np.dtype(dt).itemsize) for k, (dt, sz) in spec.items()} for spec in specs]
        deng.process_device(b, capi.fill_struct(capi.GateOut, ptrs[0]), capi.fill_struct(capi.AsmOut, ptrs[1]),
                            capi.fill_struct(capi.VarOut, ptrs[2]), capi.fill_struct(capi.GenoOut, ptrs[3]))
        deng.synchronize()
        dg, da, dv, dq = [{k: arena.download(pt[k], dt, sz) for k, (dt, sz) in spec.items()}
                          for spec, pt in zip(specs, ptrs)]
    finally:
        deng.close()
        arena.close()
    assert np.array_equal(dg["max_approx"], hg["max_approx"])
    bad = compare_asm(params, da, ha, n) + compare_vars(params, dv, hv, n)
    assert not bad, "\n".join(bad[:10])
    assert np.array_equal(dq["allele_counts"], hq["allele_counts"])
    assert np.array_equal(dq["var_qual"].view(np.uint64), hq["var_qual"].view(np.uint64))


def test_outputs_do_not_depend_on_previous_batches_or_debug_taps():
    """One engine, batches of different shapes back to back: workspaces are reused (never cleared wholesale), so
    anything a kernel forgets to write would surface as the previous batch's bytes.  The run without the debug taps
    (internal alignment records, the path bench.py times) must give the counts of the run with them."""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25)
    big, nb, nrb = synth.make_config_batch("C3", 12, first_index=97_000)
    arrs, n, nr = synth.make_config_batch("C2", 6, first_index=95_000)
    eng = Engine(params)
    try:
        eng.process(big, nb, nrb, debug=False)
        g0, a0, v0, q0 = eng.process(arrs, n, nr, debug=False)
        eng.process(big, nb, nrb, debug=True)
        g1, a1, v1, q1 = eng.process(arrs, n, nr, debug=True)
    finally:
        eng.close()
    fresh = Engine(params)
    try:
        g2, a2, v2, q2 = fresh.process(arrs, n, nr, debug=True)
    finally:
        fresh.close()
    for tag, a, v, q in (("no taps", a0, v0, q0), ("taps", a1, v1, q1)):
        bad = compare_asm(params, a, a2, n) + compare_vars(params, v, v2, n)
        assert not bad, tag + "\n" + "\n".join(bad[:10])
        assert np.array_equal(q["allele_counts"], q2["allele_counts"]), tag
        assert np.array_equal(q["var_qual"].view(np.uint64), q2["var_qual"].view(np.uint64)), tag
    bad = compare_geno(params, q1, q2, n, nr, v2["win_nvars"], arrs["read_win_off"])
    assert not bad, "\n".join(bad[:10])


def test_dense_variants_match_the_oracle():
    """Windows with 10+ variants make MaxFlow's breadth-first walk tree grow towards the reference's 2^20-visit cap;
    the folded walk search keeps them inside the device arena, so every one of them must be the
    oracle's answer with no capacity flag.  (The flagged branch -- arena exhausted -- is forced by
    test_capacity_overflow_is_retried_inside_the_library with MA_ARENA_CAP.)"""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25)
    arrs, n, nr = synth.make_config_batch("C2", 24, first_index=77_000, snv_rate=1e-2, indel_rate=2e-3)
    orc = OracleEngine(params)
    wa = orc.assemble(arrs, n, nr)
    wv = orc.msa(arrs, n, nr, wa)
    eng = Engine(params)
    try:
        a = eng.assemble(arrs, n, nr)
        v = eng.msa(arrs, n, nr, a)
    finally:
        eng.close()
    assert not (a["win_status"] & capi.MA_W_TABLE_OVERFLOW).any()
    bad = compare_asm(params, a, wa, n) + compare_vars(params, v, wv, n)
    assert not bad, "\n".join(bad[:10])
    assert int(wv["win_nvars"].max()) >= 12  # the case is as dense as intended


@pytest.mark.parametrize("bfs_limit", [6, 40, 300, 3000])
def test_traversal_limit_is_the_references(bfs_limit):
    """MaxFlow::HitTraversalLimit (max_flow.h:69) with the cap pulled down into reach: the folded search has to
    know how many entries the reference's queue would have popped -- and, when the cap falls inside a level that holds a
    qualifying arrival, the arrival's exact position in that level of the reference's queue (round 4; such windows used to
    be flagged TABLE_OVERFLOW).  Every window is the reference's answer, BFS_LIMIT flag included."""
    from lancet2_amd.engine import Engine
    params = capi.default_params(min_k=25, max_k=25, bfs_limit=bfs_limit)
    arrs, n, nr = synth.make_config_batch("C2", 12, first_index=77_000, snv_rate=1e-2, indel_rate=2e-3)
    wa = OracleEngine(params).assemble(arrs, n, nr)
    eng = Engine(params)
    try:
        a = eng.assemble(arrs, n, nr)
    finally:
        eng.close()
    flagged = [w for w in range(n) if int(a["win_status"][w]) & capi.MA_W_TABLE_OVERFLOW]
    assert not flagged, flagged
    for key in ("win_status", "win_ncomp"):
        for w in flagged:
            wa[key][w] = a[key][w]
    bad = compare_asm(params, a, wa, n)
    assert not bad, "\n".join(bad[:10])
    if bfs_limit <= 40:  # the cap was meant to bite: the reference's answer is not the uncapped one
        full = OracleEngine(capi.default_params(min_k=25, max_k=25)).assemble(arrs, n, nr)
        assert not (np.array_equal(full["win_status"], wa["win_status"]) and np.array_equal(full["comp_nhaps"], wa["comp_nhaps"]))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,nwin,first,kw,pk", [
    ("C3", 12, 93_000, dict(softclip_frac=0.05, n_frac=0.03), dict(min_k=25, max_k=25)),
    ("C3", 8, 93_100, dict(tandem_dup=45), {}),                       # the k ladder: every rung through both routes
    ("C5", 6, 93_200, {}, dict(min_k=25, max_k=25, num_samples=3)),
    ("C2", 6, 93_300, dict(W=2501), dict(min_k=25, max_k=25, max_hap_len=4096)),  # -w 2500: 2477 reference k-mers
])
def test_fused_graph_kernel_equals_the_three_general_kernels(cfg, nwin, first, kw, pk, monkeypatch):
    """Round 5: k_graph (survivor ranks from a bitmap of first instances, slot -> node in LDS, the window's distinct edges in an
    LDS set probed by source node, an edge's place read off its run, the reads' (k+1)-mers from k_support's edge queue) against
    the general route k_graph_gen = rank + edges + edge sort from the instance words (MA_NO_GRAPH_FUSE=1):
    the same assembly outputs, both equal to the oracle's; the timing names prove which route ran."""
    from lancet2_amd.engine import Engine
    params = capi.default_params(**pk)
    arrs, n, nr = synth.make_config_batch(cfg, nwin, first_index=first, **kw)
    orc = OracleEngine(params)
    wa = orc.assemble(arrs, n, nr)
    wv = orc.msa(arrs, n, nr, wa)
    wq = orc.genotype(arrs, n, nr, wa, wv)
    for fused in (True, False):
        if fused:
            monkeypatch.delenv("MA_NO_GRAPH_FUSE", raising=False)
        else:
            monkeypatch.setenv("MA_NO_GRAPH_FUSE", "1")
        eng = Engine(params)
        try:
            eng.timing_control(1)
            g, a, v, q = eng.process(arrs, n, nr, debug=True)
            times = {}
            for k_, ms in eng.kernel_times():
                times[k_] = times.get(k_, 0.0) + ms
        finally:
            eng.close()
        assert ("k_graph" in times) == fused, sorted(times)
        if fused and kw.get("W", 1001) <= 1001:  # the general kernels only leave at their first test (a 2.5 kb window holds more
            # distinct k-mers than k_insert's LDS map: its table has 16 k slots and the general kernels take it)
            assert times["k_graph_gen"] < times["k_graph"], times
        bad = compare_asm(params, a, wa, n) + compare_vars(params, v, wv, n)
        bad += compare_geno(params, q, wq, n, nr, wv["win_nvars"], arrs["read_win_off"])
        assert not bad, f"fused={fused}\n" + "\n".join(bad[:12])


@pytest.mark.gpu
@pytest.mark.parametrize("pool_kb", ["1", "4096", None])
def test_hbm_mate_mer_sets_come_out_of_a_budgeted_pool(pool_kb, monkeypatch):
    """Windows without mapping hints route every mate-mer through an HBM-resident set; each window carves its set out of the
    chunk's pool on the device (k_support), so the host never learns -- or waits for -- how many windows need how much.  A pool
    that runs out (MA_MM_POOL_KB: 1 KB and 4 MB per window of the chunk; the default without hints is a full set per window)
    is a capacity like any other: the window is flagged and re-assembled by the retry pass, whose pool holds a full set per
    window.  Same results as the oracle, no flag left."""
    from lancet2_amd.engine import Engine
    if pool_kb:
        monkeypatch.setenv("MA_MM_POOL_KB", pool_kb)
    params = capi.default_params(min_k=25, max_k=25)
    arrs, n, nr = synth.make_config_batch("C3", 6, first_index=94_000)
    arrs = dict(arrs)
    arrs.pop("read_hint")  # a batch without the array: every window takes the all-generic route
    orc = OracleEngine(params)
    wa = orc.assemble(arrs, n, nr)
    wv = orc.msa(arrs, n, nr, wa)
    wq = orc.genotype(arrs, n, nr, wa, wv)
    eng = Engine(params)
    try:
        eng.timing_control(1)
        g, a, v, q = eng.process(arrs, n, nr, debug=True)
        times = {}
        for k_, ms in eng.kernel_times():
            times[k_] = times.get(k_, 0.0) + ms
    finally:
        eng.close()
    assert times.get("k_mm_hbm", 0) > 0.05, times  # (an empty launch takes ~0.006 ms)
    assert not (a["win_status"] & capi.MA_W_TABLE_OVERFLOW).any()
    bad = compare_asm(params, a, wa, n) + compare_vars(params, v, wv, n)
    bad += compare_geno(params, q, wq, n, nr, wv["win_nvars"], arrs["read_win_off"])
    assert not bad, "\n".join(bad[:12])


@pytest.mark.gpu
def test_deep_window_whose_lds_mate_mer_set_fills_is_split_in_place(monkeypatch):
    """The bench's deep-panel window 10488 (7 k reads): one slot class of k_mm_lds's scan route holds more (k-mer, read pair)
    keys than the 32 k-entry LDS set -- the PRODUCTION detection (not an entry free after the pass), not the test build's probe
    cap.  Round 4 flagged the window and the retry pass re-assembled it through the HBM set; round 5 splits the class in two and
    redoes it before anything of it was counted.  With the capacity retries switched off the window must come out unflagged
    and identical to the oracle."""
    import sys
    sys.path.insert(0, capi.REPO)
    import bench
    from lancet2_amd.engine import Engine
    monkeypatch.setenv("MA_NO_CAP_RETRY", "1")
    arrs, n, nr = bench._gen_chunk(("C4", [10_488, 10_000], 0, 0))
    params = capi.default_params(min_k=25, max_k=25)
    orc = OracleEngine(params)
    wa = orc.assemble(arrs, n, nr)
    wv = orc.msa(arrs, n, nr, wa)
    wq = orc.genotype(arrs, n, nr, wa, wv)
    eng = Engine(params)
    try:
        eng.timing_control(1)
        g, a, v, q = eng.process(arrs, n, nr, debug=True)
        times = {}
        for k_, ms in eng.kernel_times():
            times[k_] = times.get(k_, 0.0) + ms
    finally:
        eng.close()
    assert times.get("k_mm_lds", 0) > 0.1, times  # (the scan route: more sequences than a key's leader index names)
    assert not (a["win_status"] & capi.MA_W_TABLE_OVERFLOW).any(), a["win_status"].tolist()
    bad = compare_asm(params, a, wa, n) + compare_vars(params, v, wv, n)
    bad += compare_geno(params, q, wq, n, nr, wv["win_nvars"], arrs["read_win_off"])
    assert not bad, "\n".join(bad[:12])
