cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
MA_LIB=$PWD/lancet2_amd/libmicroasm_r4.so python - > gpurun_out/r5b/dbg.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, 'tests')
import numpy as np
from harness import OracleEngine, compare_asm
from lancet2_amd import capi, synth
from lancet2_amd.engine import Engine
for ml in (2048, 4096):
    params = capi.default_params(min_k=25, max_k=25, max_hap_len=ml)
    arrs, n, nr = synth.make_config_batch("C2", 6, first_index=93_300, W=2501)
    wa = OracleEngine(params).assemble(arrs, n, nr)
    eng = Engine(params)
    a = eng.assemble(arrs, n, nr)
    eng.close()
    print(ml, capi.LIB_PATH, compare_asm(params, a, wa, n)[:3], a["win_status"].tolist(), wa["win_status"].tolist())
PY
timeout 600 python -m pytest tests -m gpu -x -q -k "fused_graph or budgeted_pool" 2>&1 | tail -5 > gpurun_out/r5b/gpu_tests2.txt
