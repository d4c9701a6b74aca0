"""GPU parity tests: the HIP path (through the C-ABI) against the oracle on identical seeded inputs.
Bit-exact for every integer/byte/index output; f64 statistics compared bit-for-bit as well (both sides
use the same sequential IEEE operations with FMA contraction off)."""
import numpy as np
import pytest

from harness import OracleEngine
from lancet2_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from lancet2_amd.engine import Engine
    e = Engine(capi.default_params())
    yield e
    e.close()


def _planted_repeat_batch():
    wins = []
    rng = np.random.default_rng(11)
    for i in range(6):
        w = synth.make_window(100 + i, W=1001, depths=(2, 2))
        ref = w["ref"]
        L = [0, 20, 30, 61, 130, 300][i]
        if L:
            ref[600:600 + L] = ref[100:100 + L]
            if i >= 3:  # sprinkle two mismatches inside the copy
                ref[600 + L // 3] = ord("A") if ref[600 + L // 3] != ord("A") else ord("C")
                ref[600 + 2 * L // 3] = ord("G") if ref[600 + 2 * L // 3] != ord("G") else ord("T")
        wins.append(w)
    wins.append(dict(ref=np.frombuffer(b"ACGT" * 50, dtype=np.uint8).copy(), reads=[]))     # pure STR
    wins.append(dict(ref=np.frombuffer(b"N" * 300, dtype=np.uint8).copy(), reads=[]))       # all N
    wins.append(dict(ref=np.frombuffer(b"ACGTTGCA", dtype=np.uint8).copy(), reads=[]))      # tiny
    wins.append(dict(ref=np.frombuffer(b"A", dtype=np.uint8).copy(), reads=[]))             # 1 base
    return synth.pack_batch(wins)


def test_repeat_gate_parity(engine):
    arrs, n, nr = _planted_repeat_batch()
    want = OracleEngine(engine.p).gate(arrs, n, nr)
    got = engine.gate(arrs, n, nr)
    assert np.array_equal(got["max_approx"], want["max_approx"]), (got["max_approx"], want["max_approx"])
    assert np.array_equal(got["max_exact"], want["max_exact"])


def test_repeat_gate_parity_random(engine):
    arrs, n, nr = synth.make_config_batch("C1", 8, depths=(1, 1))
    want = OracleEngine(engine.p).gate(arrs, n, nr)
    got = engine.gate(arrs, n, nr)
    assert np.array_equal(got["max_approx"], want["max_approx"])
    assert np.array_equal(got["max_exact"], want["max_exact"])
