"""CPU tests: the oracle reproduces the committed golden vectors, the synthetic generator is
deterministic, and the multi-GPU sharding path (gloo, world_size 2) covers every window exactly once."""
import glob
import os

import numpy as np
import pytest

from harness import OracleEngine, compare_asm, compare_cx, compare_geno, compare_vars
from lancet2_amd import capi, shard, synth

GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
                if not os.path.basename(p).startswith("pins_"))  # pins_*: brute-force score pins (test_aligner_pins.py)


def load_golden(path):
    z = np.load(path)
    meta = z["meta"]
    n, nr = int(meta[0]), int(meta[1])
    params = capi.Params(*[int(x) for x in meta[2:]])
    arrs = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    outs = {pref: {k[len(pref) + 1:]: z[k] for k in z.files if k.startswith(pref + "_")} for pref in ("gate", "asm", "var", "geno", "cx")}
    return params, arrs, n, nr, outs


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_matches_golden(path):
    params, arrs, n, nr, want = load_golden(path)
    orc = OracleEngine(params)
    g = orc.gate(arrs, n, nr)
    assert np.array_equal(g["max_approx"], want["gate"]["max_approx"])
    assert np.array_equal(g["max_exact"], want["gate"]["max_exact"])
    a = orc.assemble(arrs, n, nr)
    assert not compare_asm(params, a, want["asm"], n)
    v = orc.msa(arrs, n, nr, a)
    assert not compare_vars(params, v, want["var"], n)
    q = orc.genotype(arrs, n, nr, a, v)
    assert not compare_geno(params, q, want["geno"], n, nr, v["win_nvars"], arrs["read_win_off"])
    compare_cx(params, orc.annotate(arrs, n, nr, a, v), want["cx"], v["win_nvars"])


def test_golden_vectors_are_nontrivial():
    assert len(GOLDEN) >= 4
    tot_vars = 0
    for p in GOLDEN:
        _, _, n, nr, want = load_golden(p)
        tot_vars += int(want["var"]["win_nvars"].sum())
        assert (want["asm"]["win_ncomp"] > 0).any()
    assert tot_vars >= 8


def test_generator_is_deterministic():
    a1, n1, r1 = synth.make_config_batch("C1", 2, first_index=77)
    a2, n2, r2 = synth.make_config_batch("C1", 2, first_index=77)
    assert n1 == n2 and r1 == r2 and all(np.array_equal(a1[k], a2[k]) for k in a1)
    # collector order (core/read_collector.cpp:42-53): pass-filter first, then role, sample, qname
    flags = a1["read_flags"][: int(a1["read_win_off"][1])]
    passf = (flags & capi.MA_RF_PASS) > 0
    assert not (np.diff(passf.astype(int)) > 0).any()


def test_tile_batch_offsets():
    arrs, n, nr = synth.make_config_batch("C1", 2, first_index=5, depths=(4, 4))
    t, tn, tnr = synth.tile_batch(arrs, n, nr, 3)
    assert tn == 3 * n and tnr == 3 * nr
    assert t["ref_off"][-1] == 3 * arrs["ref_off"][-1] and t["read_off"][-1] == 3 * arrs["read_off"][-1]
    for rep in range(3):
        lo, hi = int(t["ref_off"][rep * n]), int(t["ref_off"][rep * n + 1])
        assert np.array_equal(t["ref_bases"][lo:hi], arrs["ref_bases"][: int(arrs["ref_off"][1])])


def test_shard_indices_cover_every_window_once():
    for n, world in ((10, 1), (10, 2), (17, 4), (3, 8)):
        seen = sorted(i for r in range(world) for i in shard.shard_indices(n, r, world))
        assert seen == list(range(n))
        per_rank = [[f"w{i}" for i in shard.shard_indices(n, r, world)] for r in range(world)]
        assert shard.merge_shards(per_rank, n, world) == [f"w{i}" for i in range(n)]


def test_shard_indices_in_blocks_keep_a_periodic_mix():
    """bench.py's window list has a difficult window of one kind at every 8th index, of another at every 16th, ...: dealt out
    one by one (i mod G) rank 7 of 8 would hold every tandem-repeat window and rank 0 none.  In blocks of the period every
    rank holds the same number of each kind; the blocks still cover every window once and merge_shards inverts them."""
    for n, world, block in ((64 * 8, 8, 32), (32 * 6, 2, 32), (100, 4, 32), (10, 2, 3)):
        shards = [shard.shard_indices(n, r, world, block) for r in range(world)]
        assert sorted(i for sh in shards for i in sh) == list(range(n))
        per_rank = [[f"w{i}" for i in sh] for sh in shards]
        assert shard.merge_shards(per_rank, n, world, block) == [f"w{i}" for i in range(n)]
    shards = [shard.shard_indices(32 * 8 * 4, r, 8, 32) for r in range(8)]
    for sh in shards:  # the same mix on every rank
        assert len(sh) == 128
        assert sum(1 for i in sh if i % 8 == 7) == 16 and sum(1 for i in sh if i % 16 == 3) == 8 and sum(1 for i in sh if i % 32 == 13) == 4
    one_by_one = shard.shard_indices(32 * 8 * 4, 7, 8)  # (what block = 1 does to such a list)
    assert all(i % 8 == 7 for i in one_by_one)


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        n = 5
        mine = shard.shard_indices(n, rank, world)
        params = capi.default_params(min_k=25, max_k=25)
        wins = [synth.make_window(3000 + i, W=600, depths=(12, 12)) for i in mine]
        arrs, nw, nr = synth.pack_batch(wins)
        asm = OracleEngine(params).assemble(arrs, nw, nr)
        ks = [int(k) for k in asm["win_k"]]
        t = shard.max_over_ranks(float(rank + 1), dist)  # max over ranks -> world
        gathered = [None] * world
        dist.all_gather_object(gathered, (mine, ks))
        q.put((rank, t, gathered))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, t, gathered in res:
        assert t == 2.0
        covered = sorted(i for mine, _ in gathered for i in mine)
        assert covered == list(range(5))
        assert all(k == 25 for _, ks in gathered for k in ks)


def test_bench_gpus_flag_starts_ranks_or_says_why_not():
    """`python bench.py --gpus N` without a launcher must start N ranks itself (the flag was parsed and never
    used).  On a box with fewer GPUs it must say so instead of silently running one rank and printing n_gpus = 1."""
    import json
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU box runs the real thing")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(capi.REPO, "bench.py"), "--gpus", "2", "--steps", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2
    msg = json.loads(r.stdout.strip().splitlines()[-1])
    assert "error" in msg and "--gpus 2" in msg["error"]


def test_bench_algorithmic_bytes_are_the_surveys():
    """bench.py prices its roofline with SURVEY.md 8(d)'s algorithmic bytes and nothing else: at the survey's own worked
    C3 example (R = 690, B = 103.5 k, S = 2, N_inst = 87.9 k, N_raw = 15 k, H = 3, L = 1 k, n_cigar = 16) the per-stage
    figures are the survey's 1.98 / 0.36 / 0.03 / 0.39 MB and the total its ~2.8 MB per window-attempt."""
    import bench
    st = dict(W=1001, R=690, B=103_500, S=2, N_inst=87_900, N_raw=15_000, H=3, L=1000, L_ref=1000, var_bases=20, n_cigar=16)
    mb = {s: bench.survey_bytes(s, st) / 1e6 for s in ("gate", "build", "clean", "poa", "genotype")}
    assert abs(mb["build"] - 1.98) < 0.02 and abs(mb["clean"] - 0.36) < 0.03
    assert abs(mb["poa"] - 0.03) < 0.005 and abs(mb["genotype"] - 0.39) < 0.01
    assert abs(sum(mb.values()) - 2.8) < 0.1


def test_concordance_tool_self_test():
    """tools/concordance.py (engine records vs a Lancet2 VCF, for when a built reference is at hand) compares call sets and
    QUALs the way it says"""
    import subprocess
    import sys as _sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "concordance.py")
    out = subprocess.check_output([_sys.executable, tool, "--self-test"], text=True)
    assert "self-test ok" in out
