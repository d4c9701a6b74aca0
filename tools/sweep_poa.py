"""Developer tool: POA / variant-extraction parity (engine vs oracle) over many more haplotype shapes than the test-suite
holds -- homopolymer and short-tandem-repeat indels (the closed-form first alignments place a sliding indel), dense
substitutions, long indels, with and without MA_POA_NO_DIRECT.  usage: python tools/sweep_poa.py [seed shift]"""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from harness import OracleEngine, compare_vars  # noqa: E402
from lancet2_amd import capi, synth  # noqa: E402
from lancet2_amd.engine import Engine  # noqa: E402

shift = int(sys.argv[1]) if len(sys.argv) > 1 else 0
CASES = [("C2", 64, dict(indel_rate=8e-4)), ("C2", 64, dict(str_unit=b"A", indel_rate=6e-4)),
         ("C2", 48, dict(str_unit=b"AC", indel_rate=6e-4)), ("C2", 48, dict(str_unit=b"T", snv_rate=3e-3)),
         ("C2", 32, dict(big_indel=30, str_unit=b"A")), ("C3", 32, dict(indel_rate=1e-3, snv_rate=2e-3)),
         ("C2", 32, dict(W=600, indel_rate=2e-3)), ("C2", 24, dict(snv_rate=8e-3))]
tot = 0
for cfg, nwin, kw in CASES:
    params = capi.default_params(min_k=25, max_k=45, k_step=10)
    arrs, n, nr = synth.make_config_batch(cfg, nwin, first_index=90_000 + shift, **kw)
    orc = OracleEngine(params)
    asm = orc.assemble(arrs, n, nr)
    want = orc.msa(arrs, n, nr, asm)
    eng = Engine(params)
    got = eng.msa(arrs, n, nr, asm)
    eng.close()
    bad = compare_vars(params, got, want, n)
    nal = int(sum(max(int(x) - 1, 0) for x in asm["comp_nhaps"].reshape(-1)))
    print(cfg, nwin, kw, "alignments", nal, "variants", int(want["win_nvars"].sum()), "OK" if not bad else bad[:5], flush=True)
    tot += len(bad)
sys.exit(1 if tot else 0)
