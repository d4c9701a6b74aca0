import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from lancet2_amd import capi, synth
from lancet2_amd.engine import Engine
from pin_cases import many_bubble_window
params = capi.default_params(min_k=25, max_k=25, max_hap_len=4096)
wins = [many_bubble_window(9 + i, ns) for i, ns in enumerate((10, 30, 45, 52, 60))]
arrs, n, nr = synth.pack_batch(wins)
eng = Engine(params)
a = eng.assemble(arrs, n, nr)
print("status", a["win_status"], "ncomp", a["win_ncomp"])
eng.close()
