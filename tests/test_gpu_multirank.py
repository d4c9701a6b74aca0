"""The N > 1 launch path of bench.py, rehearsed on ONE device (no scaling claimed): `--gpus 2 --oversubscribe` starts two
ranks through the same launcher parent the driver's multi-GPU tier relies on (torch.distributed.run as a child of a
process that never touched the GPU), both ranks run on device 0 with half of the HBM planned each, gloo carries the
barrier and the max-over-ranks.  Checked: rank 0's line says n_gpus 2, the ranks ran DISJOINT shards of one seeded
window list (blocks of 32 windows, block b -> rank b mod 2, lancet2_amd/shard.py; core/pipeline_executor.cpp:174-197 is the reference's
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
    # one seeded list 10000 ... 10000 + 2047, dealt out in blocks of 32 windows (the period of the workload's difficult
    # windows: every rank gets the same mix), block b -> rank b mod 2: disjoint, interleaved, complete
    from lancet2_amd.shard import shard_indices
    want = [[10_000 + i for i in shard_indices(2048, r, 2, 32)] for r in range(2)]
    assert want[0][:33] == list(range(10_000, 10_032)) + [10_064] and sorted(want[0] + want[1]) == list(range(10_000, 12_048))
    for r in range(2):
        assert (rw[r]["first"], rw[r]["last"], rw[r]["count"]) == (want[r][0], want[r][-1], 1024)
        assert rw[r]["index_sum_mod_2_31"] == sum(want[r]) % (1 << 31)
    assert out["config"]["windows_per_step_per_gpu"] == 1024
    # the metric counts the windows BOTH ranks assembled (added up over the ranks, not rank 0's share times two)
    assert 0.6 < out["config"]["assembled_fraction"] < 0.9, out["config"]["assembled_fraction"]
    ps = out["parity_sample"]
    assert ps["windows"] == 8 and ps["mismatches"] == 0, ps
