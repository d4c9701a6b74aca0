"""Seeded input generators for the aligner pins (inputs only: no scoring, nothing from oracle/)."""
import numpy as np  # noqa: F401

BASES = b"ACGT"


def rand_dna(rng, n):
    return bytes(BASES[x] for x in rng.integers(0, 4, n))


def mutate(rng, s, sub=0.0, indels=()):
    """substitutions at rate `sub`, then the given [(kind, length)] indels at random interior positions"""
    b = bytearray(s)
    for i in range(len(b)):
        if rng.random() < sub and bytes([b[i]]) in (b"A", b"C", b"G", b"T"):
            b[i] = BASES[(BASES.index(bytes([b[i]])) + int(rng.integers(1, 4))) % 4]
    for kind, ln in indels:
        if len(b) < 2 * ln + 40:
            continue
        p = int(rng.integers(20, len(b) - ln - 20))
        if kind == "D":
            del b[p:p + ln]
        else:
            b[p:p] = rand_dna(rng, ln)
    return bytes(b)


def make_pair(rng, case):
    """one (read, haplotype) pair of the given flavour"""
    n = int(rng.integers(300, 900))
    hap = rand_dna(rng, n)
    m = int(rng.choice([101, 150, 150, 150, 250]))
    if case == "str":  # tandem repeat inside the haplotype: seeds on many diagonals
        unit = rand_dna(rng, int(rng.integers(1, 7)))
        rep = (unit * 80)[: int(rng.integers(30, 90))]
        p = int(rng.integers(100, max(n - 200, 101)))
        hap = hap[:p] + rep + hap[p + len(rep):]
    start = int(rng.integers(0, n - m))
    if case == "left":
        start = -int(rng.integers(1, m - 20))
    elif case == "right":
        start = n - int(rng.integers(20, m - 1))
    src = rand_dna(rng, 400) + hap + rand_dna(rng, 400)
    read = src[400 + start: 400 + start + m + 40]
    if case == "clean":
        read = mutate(rng, read, sub=0.0)
    elif case == "noisy":
        read = mutate(rng, read, sub=float(rng.choice([0.01, 0.03, 0.06, 0.1])))
    elif case in ("indel", "str", "left", "right"):
        k = int(rng.integers(1, 3))
        sizes = [int(rng.choice([1, 2, 3, 5, 8, 12, 19, 20, 21, 25, 40])) for _ in range(k)]
        read = mutate(rng, read, sub=float(rng.choice([0.0, 0.01, 0.03])),
                      indels=[(str(rng.choice(["I", "D"])), s) for s in sizes])
    elif case == "amb":
        b = bytearray(mutate(rng, read, sub=0.01))
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, len(b)))] = ord("N")
        read = bytes(b)
        hb = bytearray(hap)
        hb[int(rng.integers(0, n))] = ord("N")
        hap = bytes(hb)
    elif case == "unrelated":
        read = rand_dna(rng, m)
    return read[:m], hap


CASES = ["clean", "noisy", "indel", "str", "left", "right", "amb", "unrelated"]


def make_haplotypes(rng, nh, L):
    ref = rand_dna(rng, L)
    haps = [ref]
    for _ in range(nh - 1):
        k = int(rng.integers(1, 4))
        sizes = [int(rng.choice([1, 2, 5, 19, 20, 21, 22, 25, 40])) for _ in range(k)]
        haps.append(mutate(rng, ref, sub=float(rng.choice([0.0, 0.005, 0.02])),
                           indels=[(str(rng.choice(["I", "D"])), s) for s in sizes]))
    return haps


# ---- crafted windows (error-free reads drawn from explicit haplotypes) ---------------------------------------------

def window_from_haps(rng, ref, samples, depth=30, read_len=150, qname0=0):
    """samples: [(list of haplotype byte strings, role)]; paired error-free reads tiling every haplotype at
    depth / len(haplotypes); returns a synth.pack_batch window dict (collector order)"""
    reads, qn = [], qname0
    for s, (hset, role) in enumerate(samples):
        for h in hset:
            n_frag = int(depth / len(hset) * len(h) / (2 * read_len))
            for _ in range(n_frag):
                ins = int(max(read_len + 10, rng.normal(400, 50)))
                fs = int(rng.integers(0, max(1, len(h) - ins)))
                frag = h[fs:fs + ins]
                for start, seq, rev in ((fs, frag[:read_len], False), (fs + ins - read_len, frag[-read_len:], True)):
                    reads.append(dict(seq=np.frombuffer(seq, np.uint8).copy(), qual=np.full(len(seq), 35, np.uint8),
                                      qname=qn, sample=s, role=role, rev=rev, passf=True, start=start, hint=-(1 << 31)))
                qn += 1
    reads.sort(key=lambda r: (0 if r["passf"] else 1, r["role"], r["sample"], r["qname"], r["start"]))
    return dict(ref=np.frombuffer(ref, np.uint8).copy(), reads=reads)


def many_bubble_window(seed, nsites, spacing=38):
    """a window of 520 + nsites * spacing + 160 bases in two stretches of read coverage: [0, 400) with one heterozygous
    SNV, and [520, end) with `nsites` heterozygous SNVs `spacing` bases apart (every one a bubble at k < spacing).
    ~30-45 bubbles exhaust MaxFlow's 2^20-visit cap (max_flow.h:69); >= 50 trip the complexity gate
    (graph_complexity.h:112-121: cyclomatic complexity >= 50 and branch points >= 50)."""
    rng = np.random.default_rng(seed)
    W = 520 + nsites * spacing + 160
    ref = rand_dna(rng, W)
    alt = bytearray(ref)
    for p in [200] + [600 + x * spacing for x in range(nsites)]:
        alt[p] = BASES[(BASES.index(bytes([alt[p]])) + 1) % 4]
    alt = bytes(alt)
    reads = []
    for part, (r_, a_) in enumerate(((ref[:400], alt[:400]), (ref[520:], alt[520:]))):
        reads += window_from_haps(rng, r_, [([r_, a_], 0), ([r_, a_], 1)], qname0=100000 * part)["reads"]
    reads.sort(key=lambda r: (0 if r["passf"] else 1, r["role"], r["sample"], r["qname"], r["start"]))
    return dict(ref=np.frombuffer(ref, np.uint8).copy(), reads=reads)
