"""Generates tests/golden/pins_aligner.npz: seeded random (read, haplotype) pairs and the optimal score of the
canonical read<->haplotype aligner for each, computed by tests/brute_force.py ALONE (numpy; nothing under oracle/
is imported), so the fixture pins the oracle to an independent implementation.
Run from the repo root:  python tests/golden/make_aligner_pins.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import brute_force as bf  # noqa: E402
from pin_cases import CASES, make_pair  # noqa: E402  (seeded inputs only)


def main():
    rng = np.random.default_rng(20261003)
    reads, haps, hit, score, case = [], [], [], [], []
    for x in range(320):
        c = CASES[x % len(CASES)]
        r, h = make_pair(rng, c)
        ok, s = bf.canonical_pair(r, h)
        reads.append(r)
        haps.append(h)
        hit.append(ok)
        score.append(s)
        case.append(CASES.index(c))
    ro = np.cumsum([0] + [len(r) for r in reads]).astype(np.int64)
    ho = np.cumsum([0] + [len(h) for h in haps]).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "pins_aligner.npz"),
                        reads=np.frombuffer(b"".join(reads), np.uint8), haps=np.frombuffer(b"".join(haps), np.uint8),
                        read_off=ro, hap_off=ho, hit=np.array(hit), score=np.array(score, np.int32),
                        case=np.array(case, np.int8))
    print("pairs", len(hit), "hits", int(np.sum(hit)))


if __name__ == "__main__":
    main()
