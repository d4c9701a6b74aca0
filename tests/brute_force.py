"""Independent brute-force scorers that pin the two restated third-party aligners.

Optimal SCORES do not depend on an implementation's tie rules, so they can be checked against something that shares
no code and no formulation with oracle/: plain numpy, one DP row at a time, horizontal gap chains closed with a
running maximum instead of the oracle's E/F recurrences.

  * read <-> haplotype (minimap2 contract, caller/genotyper.cpp:89-191, scoring_constants.h:17-20): overlap
    alignment, +1 / -4, ambiguous -1, gap(L) = 12 + 3 L, restricted to a diagonal range (or unrestricted);
  * haplotype <-> POA graph (SPOA contract, caller/msa_builder.h:64-77): global sequence-to-DAG alignment, match 0,
    mismatch -6, gap(L) = max(-6 - 2 (L-1), -26 - (L-1)).
"""
import numpy as np

NEG = -(10 ** 9)
_CODE = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3, ord("a"): 0, ord("c"): 1, ord("g"): 2, ord("t"): 3}


def encode(seq: bytes) -> np.ndarray:
    return np.array([_CODE.get(c, 4) for c in seq], dtype=np.int64)


def seed_votes(read: bytes, hap: bytes, k: int = 11):
    """{diagonal hap_pos - read_pos: number of shared exact k-mers of A/C/G/T on it}"""
    q, t = encode(read), encode(hap)
    index = {}
    for j in range(len(t) - k + 1):
        w = t[j:j + k]
        if (w > 3).any():
            continue
        index.setdefault(bytes(w.astype(np.uint8)), []).append(j)
    votes = {}
    for i in range(len(q) - k + 1):
        w = q[i:i + k]
        if (w > 3).any():
            continue
        for j in index.get(bytes(w.astype(np.uint8)), ()):
            votes[j - i] = votes.get(j - i, 0) + 1
    return votes


def seed_diagonals(read: bytes, hap: bytes, K: int, k: int = 11, min_chain_votes: int = 4):
    """(vmin, vmax) over the ANCHOR diagonals -- seeded diagonals d whose neighbourhood [d - K, d + K] holds at least
    `min_chain_votes` votes -- or None"""
    votes = seed_votes(read, hap, k)
    anchors = [d for d in votes if sum(c for e, c in votes.items() if abs(e - d) <= K) >= min_chain_votes]
    return (min(anchors), max(anchors)) if anchors else None


def reach(m: int, min_score: int = 80) -> int:
    c = m - min_score - 12
    return c // 3 if c > 0 else 0


def overlap_best(read: bytes, hap: bytes, lo: int = None, hi: int = None, go: int = 12, ge: int = 3):
    """Best score of an overlap alignment of `read` (rows) against `hap` (columns): free start on row 0 and column 0,
    free end on the last row and the last column, cells restricted to diagonals lo <= j - i <= hi (None: everywhere).
    Returns (score, i_end, j_end) with the oracle's end-cell tie rule (larger i, then smaller j) -- the score is what
    the pin compares; the end cell is a convenience."""
    q, t = encode(read), encode(hap)
    m, n = len(q), len(t)
    lo = -m if lo is None else lo
    hi = n if hi is None else hi
    cols = np.arange(n + 1)

    def allowed(i):
        d = cols - i
        return (d >= lo) & (d <= hi)

    H = np.where(allowed(0), 0, NEG).astype(np.int64)          # row 0: free start
    F = np.full(n + 1, NEG, dtype=np.int64)
    best = (NEG, -1, -1)

    def consider(score, i, j):
        nonlocal best
        if score > best[0] or (score == best[0] and (i > best[1] or (i == best[1] and j < best[2]))):
            best = (int(score), i, j)

    if allowed(0)[n] and m == 0:
        consider(0, 0, n)
    for i in range(1, m + 1):
        ok = allowed(i)
        sub = np.where((t > 3) | (q[i - 1] > 3), -1, np.where(t == q[i - 1], 1, -4))   # column j pairs hap[j-1]
        Fn = np.maximum(H - (go + ge), F - ge)                  # vertical gap: from the row above, same column
        diag = np.full(n + 1, NEG, dtype=np.int64)
        diag[1:] = H[:-1] + sub
        base = np.maximum(diag, Fn)                             # best way into the cell that is not a horizontal gap
        base[0] = 0                                             # column 0: free start (read overhangs the left end)
        base = np.where(ok, base, NEG)
        Fn = np.where(ok, Fn, NEG)
        Fn[0] = NEG
        # horizontal gaps: E(j) = max_{k<j} base(k) - go - ge (j - k), every cell between k and j allowed (the allowed
        # cells of a row are one interval, and a disallowed base is NEG, so a running maximum is enough)
        g = base + ge * cols
        run = np.maximum.accumulate(g)
        E = np.full(n + 1, NEG, dtype=np.int64)
        E[1:] = run[:-1] - go - ge * cols[1:]
        E = np.where(ok, E, NEG)
        Hn = np.maximum(base, E)
        Hn = np.where(Hn < NEG // 2, NEG, Hn)
        H, F = Hn, np.where(Fn < NEG // 2, NEG, Fn)
        if i < m and ok[n] and H[n] > NEG // 2:
            consider(H[n], i, n)
    for j in range(n + 1):
        if H[j] > NEG // 2:
            consider(H[j], m, j)
    return best


def canonical_pair(read: bytes, hap: bytes, min_score: int = 80):
    """(hit, score) of the canonical read<->haplotype aligner as DESIGN.md defines it, by brute force."""
    K = reach(len(read), min_score)
    sd = seed_diagonals(read, hap, K)
    if sd is None:
        return False, 0
    score, _, _ = overlap_best(read, hap, sd[0] - K, sd[1] + K)
    return (score >= min_score), (score if score >= min_score else 0)


def cigar_score(read: bytes, hap: bytes, rs: int, cigar):
    """Score of an alignment given as [(op, len)] (S I D M) starting at haplotype position rs: what the DP claims."""
    q, t = encode(read), encode(hap)
    qp, tp, score = 0, rs, 0
    for op, ln in cigar:
        if op == "S":
            qp += ln
        elif op == "M":
            for _ in range(ln):
                a, b = q[qp], t[tp]
                score += -1 if (a > 3 or b > 3) else (1 if a == b else -4)
                qp += 1
                tp += 1
        elif op == "I":
            score -= 12 + 3 * ln
            qp += ln
        elif op == "D":
            score -= 12 + 3 * ln
            tp += ln
    return score, qp, tp


# ---- sequence-to-DAG global alignment with the convex gap model -------------------------------------------------

def convex_gap(L, g1=-6, e1=-2, g2=-26, e2=-1):
    """SPOA kConvex: the better of two affine models, gap(L) = max(g1 + (L-1) e1, g2 + (L-1) e2) (msa_builder.h:64-77)."""
    return max(g1 + (L - 1) * e1, g2 + (L - 1) * e2)


def dag_global_score(seq: bytes, node_base, preds, order, sinks, match=0, mismatch=-6, gaps=((-6, -2), (-26, -1))):
    """Optimal score of a global alignment (SPOA kNW) of `seq` against a DAG: node_base[v] letter of node v, preds[v]
    its predecessor list (empty: source), `order` a topological order, `sinks` the nodes without successors.  Gaps in
    either direction cost the maximum over the affine models in `gaps`, each model tracked on its own (a gap run uses
    ONE model from its first to its last base -- that is what makes the model convex and not piecewise).  Plain
    O(V * |seq| * models) numpy; graph gaps (a node skipped) extend through ANY predecessor."""
    s = np.frombuffer(seq, dtype=np.uint8)
    L = len(s)
    nm = len(gaps)
    cols = np.arange(L + 1)

    def seq_gap_row(base):
        """given `base` (best non-horizontal entry per column) close horizontal gap chains for every model"""
        outs = []
        for (g, e) in gaps:
            # E(j) = max_{k<j} base(k) + g + (j-k-1) e
            run = np.maximum.accumulate(base - e * cols)
            E = np.full(L + 1, NEG, dtype=np.int64)
            E[1:] = run[:-1] + g + e * (cols[1:] - 1)
            outs.append(E)
        return outs

    # virtual start row: before any node; only horizontal gaps
    H0 = np.full(L + 1, NEG, dtype=np.int64)
    H0[0] = 0
    for E in seq_gap_row(H0):
        H0 = np.maximum(H0, E)
    H0[0] = 0
    Hrow, Frow = {}, {}
    for v in order:
        ps = preds[v]
        prevH = [H0] if not ps else [Hrow[p] for p in ps]
        prevF = [[np.full(L + 1, NEG, dtype=np.int64)] * nm] if not ps else [Frow[p] for p in ps]
        sub = np.where(s == node_base[v], match, mismatch).astype(np.int64)
        diag = np.full(L + 1, NEG, dtype=np.int64)
        Fm = [np.full(L + 1, NEG, dtype=np.int64) for _ in range(nm)]
        for ph, pf in zip(prevH, prevF):
            d = np.full(L + 1, NEG, dtype=np.int64)
            d[1:] = ph[:-1] + sub
            diag = np.maximum(diag, d)
            for x, (g, e) in enumerate(gaps):   # node v is skipped by the sequence: vertical gap, per model
                Fm[x] = np.maximum(Fm[x], np.maximum(ph + g, pf[x] + e))
        base = diag
        for x in range(nm):
            base = np.maximum(base, Fm[x])
        Hv = base
        for E in seq_gap_row(base):
            Hv = np.maximum(Hv, E)
        Hrow[v] = np.where(Hv < NEG // 2, NEG, Hv)
        Frow[v] = [np.where(f < NEG // 2, NEG, f) for f in Fm]
    return int(max(Hrow[v][L] for v in sinks))
