"""Developer tool: per-kernel HIP-event times of one ma_process_batch call (host buffers)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lancet2_amd import capi, synth  # noqa: E402
from lancet2_amd import engine as E  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
arrs, nw, nr = synth.make_config_batch("C2", 64, first_index=first)
arrs, nw, nr = synth.tile_batch(arrs, nw, nr, n // 64)
eng = E.Engine(capi.default_params(min_k=25, max_k=25))
eng.process(arrs, nw, nr)
eng.timing_control(1)
eng.process(arrs, nw, nr)
agg = {}
for k, v in eng.kernel_times():
    agg[k] = agg.get(k, 0.0) + v
print({k: round(v, 2) for k, v in sorted(agg.items(), key=lambda kv: -kv[1])})
eng.close()
