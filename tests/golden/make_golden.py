#!/usr/bin/env python3
"""Generates tests/golden/*.npz: seeded synthetic windows (inputs) + the oracle's outputs for every stage.

The reference itself cannot be built or imported here, so these vectors pin the ORACLE
(which is in turn pinned to the reference's known-answer tests by tests/test_oracle_kat.py); they guard
both the oracle against regressions (CPU test) and the HIP path against the oracle (GPU test).
Run from the repo root:  python tests/golden/make_golden.py [case ...]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from harness import OracleEngine  # noqa: E402
from lancet2_amd import capi, synth  # noqa: E402

CASES = {
    # name: (config, n_windows, first_index, param overrides, generator overrides)
    "c1_k25": ("C1", 2, 1000, dict(min_k=25, max_k=25), {}),
    "c2_cascade": ("C2", 2, 1100, {}, {}),
    "c5_three_samples": ("C5", 1, 1200, dict(min_k=25, max_k=25, num_samples=3), {}),
    "c4_indel50": ("C4", 1, 1300, dict(min_k=25, max_k=25), dict(depths=(40, 40))),
    # 18 variants / 5 haplotypes in one window: MaxFlow's walk tree is 2^18 prefixes wide in the reference's search
    "c2_dense_variants": ("C2", 1, 77_009, dict(min_k=25, max_k=25), dict(snv_rate=1e-2, indel_rate=2e-3)),
}


def main():
    only = set(sys.argv[1:])  # optional: regenerate just the named cases
    for name, (cfg, n, first, pk, gk) in CASES.items():
        if only and name not in only:
            continue
        params = capi.default_params(**pk)
        arrs, nw, nr = synth.make_config_batch(cfg, n, first_index=first, **gk)
        orc = OracleEngine(params)
        out = {}
        out.update({"in_" + k: v for k, v in arrs.items()})
        g = orc.gate(arrs, nw, nr)
        a = orc.assemble(arrs, nw, nr)
        v = orc.msa(arrs, nw, nr, a)
        q = orc.genotype(arrs, nw, nr, a, v, debug=True)
        cx = orc.annotate(arrs, nw, nr, a, v)  # SEQ_CX / GRAPH_CX at the default GC fraction 0.41
        for pref, d in (("gate_", g), ("asm_", a), ("var_", v), ("geno_", q), ("cx_", cx)):
            out.update({pref + k: val for k, val in d.items()})
        out["meta"] = np.array([nw, nr] + [getattr(params, f) for f, _ in capi.Params._fields_], dtype=np.int64)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, "windows", nw, "reads", nr, "variants", int(v["win_nvars"].sum()), os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
