"""Developer tool: one-off bit-exact parity sweep of the whole chain over many more seeds than the test-suite uses."""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from harness import OracleEngine, compare_asm, compare_cx, compare_geno, compare_vars  # noqa: E402
from lancet2_amd import capi, synth  # noqa: E402
from lancet2_amd.engine import Engine  # noqa: E402

CASES = [("C2", 96, 70_000, {}, dict(min_k=25, max_k=25)), ("C3", 32, 71_000, {}, dict(min_k=25, max_k=25)),
         ("C2", 24, 72_000, dict(str_unit=b"AT", n_somatic=2), {}), ("C5", 8, 73_000, {}, dict(min_k=25, max_k=25, num_samples=3)),
         ("C2", 16, 74_000, dict(W=1500, indel_rate=1e-3), dict(min_k=25, max_k=25)),
         ("C2", 12, 75_000, dict(big_indel=40), dict(min_k=25, max_k=25)),
         # many error bubbles / tips, many branch nodes, repeats through the k cascade: the graph-cleaning corner cases
         ("C2", 24, 76_000, dict(error_scale=4.0), dict(min_k=25, max_k=25)),
         ("C2", 24, 77_000, dict(snv_rate=1e-2, indel_rate=2e-3), dict(min_k=25, max_k=25)),
         ("C3", 16, 78_000, dict(error_scale=3.0, str_unit=b"CAG"), {}),
         ("C2", 16, 79_000, dict(error_scale=2.0, snv_rate=5e-3), dict(min_k=17, max_k=41, k_step=8)),
         # round 2: germline mode (PL / GQ / QUAL), long reads (wide search regions), long indels
         ("C2", 24, 80_000, {}, dict(min_k=25, max_k=25, case_ctrl_mode=0)),
         ("C2", 12, 81_000, dict(read_len=250, big_indel=60), dict(min_k=25, max_k=25)),
         ("C3", 16, 82_000, dict(str_unit=b"AGGGTT", error_scale=2.0), dict(min_k=25, max_k=25)),
         # round 3: the bench's harder shapes -- dispersed duplications (cycles at k = 25: the k ladder), low-complexity
         # stretches, soft-clipped and N-containing reads, 2 x 250 reads
         ("C3", 16, 83_000, dict(dup_len=200), {}), ("C3", 16, 84_000, dict(dup_len=260), dict(min_k=25, max_k=25)),
         ("C3", 16, 85_000, dict(low_complexity=100), {}),
         ("C3", 16, 86_000, dict(softclip_frac=0.1, n_frac=0.05), dict(min_k=25, max_k=25)),
         ("C2", 12, 87_000, dict(read_len=250, dup_len=180, softclip_frac=0.05, n_frac=0.02), {}),
         ("C3", 16, 89_000, dict(tandem_dup=40), {}), ("C3", 16, 90_000, dict(tandem_dup=70, softclip_frac=0.03), {}),
         ("C3", 12, 88_000, dict(dup_len=150, low_complexity=60, softclip_frac=0.03, n_frac=0.01, str_unit=b"CA"), {})]
shift = int(sys.argv[1]) if len(sys.argv) > 1 else 0  # other windows of the same shapes
if len(sys.argv) > 2 and sys.argv[2] == "c4":  # the deep panel at full depth: 32 windows of 500x/500x, 50 bp indels
    CASES = [("C4", 32, 91_000, {}, dict(min_k=25, max_k=25))]
tot = 0
for cfg, nwin, first, kw, pk in CASES:
    first += shift
    params = capi.default_params(**pk)
    arrs, n, nr = synth.make_config_batch(cfg, nwin, first_index=first, **kw)
    orc = OracleEngine(params)
    wg, wa = orc.gate(arrs, n, nr), orc.assemble(arrs, n, nr)
    wv = orc.msa(arrs, n, nr, wa)
    wq = orc.genotype(arrs, n, nr, wa, wv)
    wc = orc.annotate(arrs, n, nr, wa, wv)
    eng = Engine(params)
    g, a, v, q = eng.process(arrs, n, nr, debug=True)
    cx = eng.annotate(arrs, n, nr, a, v)
    eng.close()
    bad = [] if np.array_equal(g["max_approx"], wg["max_approx"]) else ["gate"]
    bad += compare_asm(params, a, wa, n) + compare_vars(params, v, wv, n)
    bad += compare_geno(params, q, wq, n, nr, wv["win_nvars"], arrs["read_win_off"])
    compare_cx(params, cx, wc, wv["win_nvars"])
    print(cfg, nwin, kw, "variants", int(wv["win_nvars"].sum()), "OK" if not bad else bad[:5], flush=True)
    tot += len(bad)
sys.exit(1 if tot else 0)
