"""The N > 1 launch path of bench.py, rehearsed on ONE device (no scaling claimed): `--gpus 2 --oversubscribe` starts two
ranks through the same launcher parent the driver's multi-GPU tier relies on (torch.distributed.run as a child of a
process that never touched the GPU), both ranks run on device 0 with half of the HBM planned each, gloo carries the
barrier and the max-over-ranks.  Checked: rank 0's line says n_gpus 2, the ranks ran DISJOINT shards of one seeded
window list (window i -> rank i mod 2, lancet2_amd/shard.py; core/pipeline_executor.cpp:174-197 is the reference's
worker model), and the engine's outputs for a sample of rank 0's own windows equal the oracle's."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_rehearsed_on_one_device():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--oversubscribe", "--steps", "1", "--warmup", "1",
           "--no-cpu", "--no-also", "--windows", "1024", "--distinct", "1024", "--parity-windows", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=REPO, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["oversubscribed"]
    rw = out["config"]["rank_windows"]
    assert [x["rank"] for x in rw] == [0, 1]
    # one seeded list 10000 ... 10000 + 2047, window i -> rank i mod 2: disjoint, interleaved, complete
    assert (rw[0]["first"], rw[0]["last"], rw[0]["count"]) == (10_000, 10_000 + 2046, 1024)
    assert (rw[1]["first"], rw[1]["last"], rw[1]["count"]) == (10_001, 10_000 + 2047, 1024)
    assert rw[0]["index_sum_mod_2_31"] == sum(range(10_000, 10_000 + 2048, 2)) % (1 << 31)
    assert rw[1]["index_sum_mod_2_31"] == sum(range(10_001, 10_000 + 2048, 2)) % (1 << 31)
    assert out["config"]["windows_per_step_per_gpu"] == 1024
    ps = out["parity_sample"]
    assert ps["windows"] == 8 and ps["mismatches"] == 0, ps
