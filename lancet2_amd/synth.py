"""Seeded synthetic tumour/normal windows in the engine's batch layout.

Data only: a random reference window inside a longer random "genome", planted germline (phased,
heterozygous/homozygous) and somatic (tumour-only, VAF 0.1-0.5) variants, paired 150 bp reads with
position-dependent qualities and matching substitution errors, reverse-strand reads stored in
reference orientation, 5 % of reads failing the MAPQ filter, sorted with the reference collector's
comparator (core/read_collector.cpp:42-53: pass-filter desc, tag, sample, qname, start).
"""
import numpy as np

from .capi import MA_RF_CASE, MA_RF_PASS, MA_RF_REV

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
SEED0 = 0x5EED5EED5EED5EED
FLANK = 600
READ_LEN = 150


def _rand_dna(rng, n):
    return BASES[rng.integers(0, 4, size=n)]


def _apply_variants(genome, variants, with_map=False):
    """variants: list of (pos, ref_len, alt_bytes) on genome coordinates, non-overlapping.
    with_map: also return [(hap_pos, shift)] breakpoints: for hap positions >= hap_pos, ref = hap - shift."""
    out = []
    last = 0
    hap_len = 0
    shift = 0
    bps = [(0, 0)]
    for pos, rlen, alt in sorted(variants, key=lambda v: v[0]):
        if pos < last:
            continue
        out.append(genome[last:pos])
        hap_len += pos - last
        out.append(np.frombuffer(alt, dtype=np.uint8) if len(alt) else np.zeros(0, np.uint8))
        hap_len += len(alt)
        last = pos + rlen
        shift += len(alt) - rlen
        bps.append((hap_len, shift))
    out.append(genome[last:])
    seq = np.concatenate(out)
    return (seq, bps) if with_map else seq


def _hap_to_ref(bps, hap_pos):
    shift = 0
    for hp, sh in bps:
        if hap_pos >= hp:
            shift = sh
        else:
            break
    return hap_pos - shift


def make_window(index, W=1001, depths=(30, 30), roles=(0, 1), snv_rate=1e-3, indel_rate=2e-4,
                n_somatic=1, big_indel=0, str_unit=None, read_len=READ_LEN, error_scale=1.0,
                low_mapq_frac=0.05, seed0=SEED0, dup_len=0, low_complexity=0, softclip_frac=0.0, n_frac=0.0, tandem_dup=0):
    """Returns dict(ref=uint8[W], reads=list of dict) for one window.

    depths/roles: per-sample depth and role (0 = normal/CTRL, 1 = tumour/CASE).
    big_indel: if > 0, plant one somatic insertion and one germline deletion of that length (C4).
    str_unit: optional bytes; a short tandem repeat (12 copies) is spliced into the window.
    dup_len: if > 0, a dispersed duplication: that many bases of the window's first third are copied into its last third
        (a repeat of the REFERENCE longer than k: the repeat gate skips every k up to its length).
    tandem_dup: if > 0, a germline tandem duplication of that many bases in one or both sample haplotypes (the reference
        window has no repeat, so the gate lets k = 25 through; the graph of the reads has a cycle until k outgrows the
        duplication -- what sends WGS windows up the k ladder).
    low_complexity: if > 0, a stretch of that many bases drawn from a skewed two-letter alphabet.
    softclip_frac: fraction of reads whose head or tail (8-40 bases) is replaced by unrelated low-quality bases
        (adapter read-through / chimeric ends: what an aligner soft-clips).
    n_frac: fraction of reads with one to three N bases (quality 2).
    The four options draw from a random stream of their own: a window without them is unchanged.
    """
    rng = np.random.default_rng((seed0 + index) & ((1 << 63) - 1))
    G = W + 2 * FLANK
    genome = _rand_dna(rng, G)
    if str_unit is not None:
        unit = np.frombuffer(str_unit, dtype=np.uint8)
        rep = np.tile(unit, 12)
        s = FLANK + W // 2
        genome[s:s + len(rep)] = rep[: max(0, min(len(rep), G - s))]
    rng2 = np.random.default_rng(((seed0 ^ 0x0DD5EED5) + 7919 * index) & ((1 << 63) - 1))  # the harder shapes' own stream
    if dup_len > 0 and W >= 3 * dup_len:
        s1 = FLANK + int(rng2.integers(40, W // 3 - dup_len // 2))
        s2 = FLANK + int(rng2.integers(2 * W // 3 - dup_len // 2, W - dup_len - 40))
        genome[s2:s2 + dup_len] = genome[s1:s1 + dup_len]
    if low_complexity > 0:
        a, b2 = BASES[rng2.permutation(4)[:2]]
        s = FLANK + int(rng2.integers(60, max(61, W - low_complexity - 60)))
        genome[s:s + low_complexity] = np.where(rng2.random(low_complexity) < 0.8, a, b2)
    # ---- variants (genome coordinates) ----
    germ = []  # (pos, ref_len, alt, hapmask) hapmask bit0 = A, bit1 = B
    lo, hi = FLANK + 60, FLANK + W - 60
    pos = lo
    while True:
        gap = int(rng.geometric(min(0.5, snv_rate + indel_rate)))
        pos += max(gap, 35)
        if pos >= hi:
            break
        if rng.random() < snv_rate / (snv_rate + indel_rate):
            alt = BASES[(np.searchsorted(BASES, genome[pos]) + rng.integers(1, 4)) % 4]
            germ.append((pos, 1, bytes([alt]), int(rng.integers(1, 4))))
        else:
            ln = int(rng.integers(1, 6))
            if rng.random() < 0.5:
                germ.append((pos, ln, b"", int(rng.integers(1, 4))))
            else:
                germ.append((pos, 0, _rand_dna(rng, ln).tobytes(), int(rng.integers(1, 4))))
    if tandem_dup > 0:
        for _try in range(30):
            p = int(rng2.integers(lo + tandem_dup + 10, hi - 10))
            if all(abs(p - g[0]) > tandem_dup + 40 for g in germ):
                germ.append((p, 0, genome[p - tandem_dup:p].tobytes(), int(rng2.integers(1, 4))))
                break
    som = []
    used = [g[0] for g in germ]
    for _ in range(n_somatic):
        for _try in range(20):
            p = int(rng.integers(lo, hi))
            if all(abs(p - u) > 40 for u in used):
                break
        used.append(p)
        kind = rng.integers(0, 3)
        if kind == 0:
            alt = BASES[(np.searchsorted(BASES, genome[p]) + rng.integers(1, 4)) % 4]
            v = (p, 1, bytes([alt]))
        elif kind == 1:
            v = (p, int(rng.integers(1, 8)), b"")
        else:
            v = (p, 0, _rand_dna(rng, int(rng.integers(1, 8))).tobytes())
        som.append((v, int(rng.integers(0, 2)), float(rng.uniform(0.1, 0.5))))
    if big_indel > 0:
        for _try in range(50):
            p1 = int(rng.integers(lo + 50, hi - 50 - big_indel))
            if all(abs(p1 - u) > 60 + big_indel for u in used):
                break
        used.append(p1)
        som.append(((p1, 0, _rand_dna(rng, big_indel).tobytes()), 0, 0.35))
        for _try in range(50):
            p2 = int(rng.integers(lo + 50, hi - 50 - big_indel))
            if all(abs(p2 - u) > 60 + 2 * big_indel for u in used):
                germ.append((p2, big_indel, b"", 1))
                used.append(p2)
                break
    # haplotype sequences: germline A/B, plus tumour subclone variants of each
    def hap_seq(hbit, with_som):
        vs = [(p, r, a) for (p, r, a, m) in germ if m & (1 << hbit)]
        vs += [v for (v, h, _vaf) in with_som if h == hbit]
        return _apply_variants(genome, vs, with_map=True)

    hapN = [hap_seq(0, []), hap_seq(1, [])]
    reads = []
    frag_id = 0
    for s, (depth, role) in enumerate(zip(depths, roles)):
        n_frag = int(round(depth * (W + 2 * read_len) / (2.0 * read_len)))
        for _ in range(n_frag):
            hb = int(rng.integers(0, 2))
            if role == 1 and som:
                carried = [x for x in som if x[1] == hb and rng.random() < min(1.0, 2.0 * x[2])]
                hseq, bps = hap_seq(hb, carried) if carried else hapN[hb]
            else:
                hseq, bps = hapN[hb]
            ins = int(max(read_len + 10, rng.normal(400, 50)))
            # fragment start so that reads tile the window incl. its edges
            fs = int(rng.integers(FLANK - ins, FLANK + W))
            fs = max(0, min(fs, len(hseq) - ins))
            frag = hseq[fs:fs + ins]
            if len(frag) < read_len:
                continue
            mates = [(fs, frag[:read_len], False), (fs + ins - read_len, frag[ins - read_len:], True)]
            name = frag_id
            frag_id += 1
            lowq = rng.random() < low_mapq_frac
            for start, seq, rev in mates:
                # keep only reads overlapping the window (approximate hap->genome coordinates)
                if start + read_len <= FLANK or start >= FLANK + W:
                    continue  # (hap coordinates are within a few bases of genome coordinates)
                x = np.arange(read_len) / (read_len - 1.0)
                q = 37.0 - 17.0 * x * x + rng.normal(0, 1.5, read_len)
                q = np.clip(np.round(q), 2, 41).astype(np.uint8)
                if rev:  # sequencing direction is reversed for reverse-strand reads
                    q = q[::-1].copy()
                seq = seq.copy()
                perr = error_scale * np.power(10.0, -q.astype(np.float64) / 10.0)
                err = rng.random(read_len) < perr
                if err.any():
                    idx = np.nonzero(err)[0]
                    seq[idx] = BASES[(np.searchsorted(BASES, seq[idx]) + rng.integers(1, 4, len(idx))) % 4]
                if softclip_frac > 0.0 and rng2.random() < softclip_frac:
                    cl = int(rng2.integers(8, 41))
                    sl = slice(0, cl) if rng2.random() < 0.5 else slice(read_len - cl, read_len)
                    seq[sl] = _rand_dna(rng2, cl)
                    q = q.copy()
                    q[sl] = rng2.integers(2, 13, cl).astype(np.uint8)
                if n_frac > 0.0 and rng2.random() < n_frac:
                    at = rng2.integers(0, read_len, int(rng2.integers(1, 4)))
                    seq[at] = ord("N")
                    q = q.copy()
                    q[at] = 2
                reads.append(dict(seq=seq, qual=q, qname=name, sample=s, role=role, rev=rev,
                                  passf=not lowq, start=start - FLANK,
                                  hint=_hap_to_ref(bps, start) - FLANK))  # what a BAM record's POS gives
    # collector order (read_collector.cpp:42-53)
    reads.sort(key=lambda r: (0 if r["passf"] else 1, r["role"], r["sample"], r["qname"], r["start"]))
    return dict(ref=genome[FLANK:FLANK + W].copy(), reads=reads)


def pack_batch(windows):
    """list of make_window() dicts -> dict of numpy arrays in ma_batch_t layout."""
    n = len(windows)
    ref_off = np.zeros(n + 1, np.uint32)
    read_win_off = np.zeros(n + 1, np.uint32)
    for i, w in enumerate(windows):
        ref_off[i + 1] = ref_off[i] + len(w["ref"])
        read_win_off[i + 1] = read_win_off[i] + len(w["reads"])
    nr = int(read_win_off[-1])
    lens = np.fromiter((len(r["seq"]) for w in windows for r in w["reads"]), dtype=np.uint64, count=nr)
    read_off = np.zeros(nr + 1, np.uint64)
    np.cumsum(lens, out=read_off[1:])
    arrs = dict(
        ref_bases=np.concatenate([w["ref"] for w in windows]) if n else np.zeros(0, np.uint8),
        ref_off=ref_off, read_win_off=read_win_off, read_off=read_off,
        read_bases=(np.concatenate([r["seq"] for w in windows for r in w["reads"]]) if nr else np.zeros(0, np.uint8)),
        read_quals=(np.concatenate([r["qual"] for w in windows for r in w["reads"]]) if nr else np.zeros(0, np.uint8)),
        read_qname_id=np.fromiter((r["qname"] for w in windows for r in w["reads"]), dtype=np.uint32, count=nr),
        read_sample=np.fromiter((r["sample"] for w in windows for r in w["reads"]), dtype=np.uint8, count=nr),
        read_flags=np.fromiter(((MA_RF_PASS if r["passf"] else 0) | (MA_RF_CASE if r["role"] == 1 else 0) |
                                (MA_RF_REV if r["rev"] else 0) for w in windows for r in w["reads"]),
                               dtype=np.uint8, count=nr),
        read_hint=np.fromiter((r.get("hint", -(1 << 31)) for w in windows for r in w["reads"]),
                              dtype=np.int32, count=nr),
    )
    # pad byte arrays so that vector loads past the end stay inside the allocation
    for k in ("ref_bases", "read_bases", "read_quals"):
        arrs[k] = np.concatenate([arrs[k], np.zeros(64, np.uint8)])
    return arrs, n, nr


def tile_batch(arrs, n, nr, times):
    """Repeat a packed batch `times` times (distinct memory, same content) for large benches."""
    if times <= 1:
        return arrs, n, nr
    out = {}
    ref = arrs["ref_bases"][:-64]
    rb = arrs["read_bases"][:-64]
    rq = arrs["read_quals"][:-64]
    out["ref_bases"] = np.concatenate([np.tile(ref, times), np.zeros(64, np.uint8)])
    out["read_bases"] = np.concatenate([np.tile(rb, times), np.zeros(64, np.uint8)])
    out["read_quals"] = np.concatenate([np.tile(rq, times), np.zeros(64, np.uint8)])
    def tile_off(off, dtype):
        base = off[:-1].astype(np.uint64)
        total = np.uint64(off[-1])
        parts = [base + np.uint64(t) * total for t in range(times)]
        return np.concatenate(parts + [np.array([total * np.uint64(times)], np.uint64)]).astype(dtype)
    out["ref_off"] = tile_off(arrs["ref_off"], np.uint32)
    out["read_win_off"] = tile_off(arrs["read_win_off"], np.uint32)
    out["read_off"] = tile_off(arrs["read_off"], np.uint64)
    for k in ("read_qname_id", "read_sample", "read_flags", "read_hint"):
        if k in arrs:
            out[k] = np.tile(arrs[k], times)
    return out, n * times, nr * times


def slice_batch(arrs, n, w0, w1):
    """windows [w0, w1) of a packed batch as a packed batch of their own"""
    w1 = min(w1, n)
    r0, r1 = int(arrs["read_win_off"][w0]), int(arrs["read_win_off"][w1])
    b0, b1 = int(arrs["read_off"][r0]), int(arrs["read_off"][r1])
    f0, f1 = int(arrs["ref_off"][w0]), int(arrs["ref_off"][w1])
    pad = np.zeros(64, np.uint8)
    out = dict(ref_bases=np.concatenate([arrs["ref_bases"][f0:f1], pad]),
               read_bases=np.concatenate([arrs["read_bases"][b0:b1], pad]),
               read_quals=np.concatenate([arrs["read_quals"][b0:b1], pad]),
               ref_off=(arrs["ref_off"][w0:w1 + 1] - arrs["ref_off"][w0]).astype(np.uint32),
               read_win_off=(arrs["read_win_off"][w0:w1 + 1] - arrs["read_win_off"][w0]).astype(np.uint32),
               read_off=(arrs["read_off"][r0:r1 + 1] - arrs["read_off"][r0]).astype(np.uint64))
    for k in ("read_qname_id", "read_sample", "read_flags", "read_hint"):
        if k in arrs:
            out[k] = arrs[k][r0:r1].copy()
    return out, w1 - w0, r1 - r0


CONFIGS = {
    # BASELINE.json configs -> generator arguments
    "C1": dict(W=600, depths=(30, 30), roles=(0, 1)),
    "C2": dict(W=1001, depths=(30, 30), roles=(0, 1)),
    "C3": dict(W=1001, depths=(30, 60), roles=(0, 1)),          # 60x tumour / 30x normal
    "C4": dict(W=1001, depths=(500, 500), roles=(0, 1), big_indel=50),
    "C5": dict(W=1001, depths=(30, 30, 30), roles=(0, 0, 1)),   # 2 normals + 1 tumour
}


def make_config_batch(cfg, n_windows, first_index=0, **over):
    kw = dict(CONFIGS[cfg])
    kw.update(over)
    wins = [make_window(first_index + i, **kw) for i in range(n_windows)]
    return pack_batch(wins)
