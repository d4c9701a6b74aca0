#!/usr/bin/env python3
"""bench.py -- microassembly windows/sec on MI355X (BASELINE.json metric).

A step = one pass of the whole hot path (repeat gate -> cbdg assembly -> POA/variants -> read<->haplotype
genotyping: ma_process_batch) over one batch of synthetic tumour/normal windows that is already resident
in HBM.  N > 1: one process per GPU (torch.distributed / RCCL only for the barrier + max-over-ranks),
windows statically sharded, NO data-path collective (windows never communicate) -> weak scaling.

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the fields).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--windows", type=int, default=2048, help="windows per step per GPU")
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic windows (tiled to --windows)")
    ap.add_argument("--config", default="C2", help="BASELINE.json config: C2 = chr22-shaped 30x/30x, k=25")
    ap.add_argument("--cpu-windows", type=int, default=24, help="oracle sample for cpu_baseline (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    return ap.parse_args()


def algorithmic_bytes(kernel, st):
    """Algorithmic HBM bytes per LAUNCH of `kernel` (SURVEY.md 8d per-window figures x windows per launch;
    inputs read once, outputs written once, intermediates counted only where they must live in HBM)."""
    n, S = st["windows_per_launch"], st["S"]
    W, R, Bb, Ni, Nraw, Nn = st["W"], st["R"], st["B"], st["N_inst"], st["N_raw"], st["N_nodes"]
    H, L, P = st["H"], st["L"], st["pairs"]
    per = {
        "gate_kernel": W + 8,
        "k_build_insert": 2 * Bb + 5 * R + W + 16 * Ni,                    # bases+quals, read meta, ref, key probe + slot/first update
        "k_mm_insert": 4 * Ni + 12 * Ni,                                   # instance word + mate-mer entry
        "k_count": 4 * Ni + 12 * Ni + 8 * Ni,
        "k_rank": 4 * Ni + (16 + 4 * S) * Nraw,                            # instance words + node records
        "k_edges": 8 * Ni,
        "k_edge_sort": 128 * Nn,
        "k_clean": (16 + 4 * S) * Nn + 64 * Nn + 5 * H * L + 64,           # node records + edge lists in, haplotypes out
        "k_msa": 5 * H * L + 16 * (L + 64),
        "k_hap_index": H * L + 10 * H * L,
        "k_vote": P * (150 + 2 * 140),
        "k_align": P * (150 + 300 + 24 + 16) ,                              # read + haplotype segment + result
        "k_assign": R * (300 + H * 40),
        "k_evidence": R * 80,
        "k_qual": 64,
        "k_plan": R * 8,
    }
    return per.get(kernel, 0) * n


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print(json.dumps({"error": "no GPU: the engine has no CPU fallback"}))
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    from lancet2_amd import capi, synth
    from lancet2_amd.engine import Engine

    params = capi.default_params(min_k=25, max_k=25)  # BASELINE config: k = 25 (single attempt)
    if args.config == "C5":
        params.num_samples = 3
    # ---- synthetic batch (distinct seeds per rank), tiled up to --windows ----
    distinct = min(args.distinct, args.windows)
    arrs, n0, nr0 = synth.make_config_batch(args.config, distinct, first_index=10_000 + rank * 100_000)
    times = max(1, args.windows // n0)
    arrs, n, nr = synth.tile_batch(arrs, n0, nr0, times)
    dbatch = {k: torch.from_numpy(v.view(np.uint8) if v.dtype != np.uint8 else v).to(dev) for k, v in arrs.items()}
    b = capi.make_batch_struct(dbatch, n, nr)

    def dev_alloc(spec):
        return {k: torch.zeros(int(sz) * np.dtype(dt).itemsize, dtype=torch.uint8, device=dev) for k, (dt, sz) in spec.items()}

    g = dev_alloc(capi.gate_out_spec(n))
    a = dev_alloc(capi.asm_out_spec(params, n))
    v = dev_alloc(capi.var_out_spec(params, n))
    q = dev_alloc(capi.geno_out_spec(params, n, nr, debug=False))
    gs, as_, vs, qs = (capi.fill_struct(capi.GateOut, g), capi.fill_struct(capi.AsmOut, a),
                       capi.fill_struct(capi.VarOut, v), capi.fill_struct(capi.GenoOut, q))
    eng = Engine(params, device=local_rank, memspace=capi.MA_MEM_DEVICE)
    stream = torch.cuda.current_stream(dev)
    eng.set_stream(stream.cuda_stream)

    def step():
        eng.process_device(b, gs, as_, vs, qs)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    eng.timing_control(0)
    for _ in range(args.warmup):
        step()
    barrier()
    eng.timing_control(2)  # HIP events around every kernel, accumulated over the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    ktimes = eng.kernel_times()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- workload statistics from the results (for the algorithmic-bytes model) ----
    status = a["win_status"].view(torch.int32).cpu().numpy().view(np.uint32)
    ncomp = a["win_ncomp"].view(torch.int32).cpu().numpy()
    nvars = v["win_nvars"].view(torch.int32).cpu().numpy()
    hap_len = a["hap_len"].view(torch.int32).cpu().numpy().reshape(n, params.max_haps)
    nhaps = a["comp_nhaps"].view(torch.int32).cpu().numpy().reshape(n, params.max_comps)
    assembled = int(((status & capi.MA_W_NO_HAPLOTYPE) == 0).sum())
    overflowed = int((status & (capi.MA_W_HAP_OVERFLOW | capi.MA_W_LEN_OVERFLOW | capi.MA_W_TABLE_OVERFLOW |
                                capi.MA_W_VAR_OVERFLOW)).astype(bool).sum())
    R = nr / n
    Bb = float(arrs["read_off"][-1]) / n
    W = float(arrs["ref_off"][-1]) / n
    k = 25
    pass_frac = float((arrs["read_flags"] & capi.MA_RF_PASS).astype(bool).mean())
    Ni = (W - k + 1) + R * pass_frac * (Bb / max(R, 1) - k + 1)
    H = float(nhaps.sum()) / max(assembled, 1)
    L = float(hap_len.sum()) / max(float((hap_len > 0).sum()), 1.0)
    st = dict(windows_per_launch=n, S=params.num_samples, W=W, R=R, B=Bb, N_inst=Ni, N_raw=0.15 * Ni,
              N_nodes=1.4 * W, H=H, L=L, pairs=R * H * (nvars > 0).mean())

    total_windows = n * args.steps * world
    wps = total_windows / elapsed
    asm_wps = assembled * args.steps * world / elapsed

    # ---- dominant kernel + roofline ----
    agg = {}
    for name, ms in ktimes:
        s = agg.setdefault(name, [0.0, 0])
        s[0] += ms
        s[1] += 1
    dom = max(agg.items(), key=lambda kv: kv[1][0])[0] if agg else None
    roof = None
    if dom:
        tot_ms, launches = agg[dom]
        avg_ms = tot_ms / launches
        launches_per_step = launches / args.steps
        bytes_per_launch = algorithmic_bytes(dom, st) / max(launches_per_step, 1)
        ach = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None,
                "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches_per_step,
                "algorithmic_bytes_per_launch": int(bytes_per_launch)}
    kernel_ms_per_step = {kname: round(val[0] / args.steps, 3) for kname, val in sorted(agg.items(), key=lambda kv: -kv[1][0])}

    # ---- CPU baseline: the oracle (a port of the reference path) on a bounded sample, rank 0, N = 1 ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu and args.cpu_windows > 0:
        sys.path.insert(0, os.path.join(REPO, "tests"))
        from harness import OracleEngine
        m = min(args.cpu_windows, n0)
        sub, sn, snr = synth.make_config_batch(args.config, m, first_index=10_000)
        orc = OracleEngine(params)
        c0 = time.perf_counter()
        og = orc.gate(sub, sn, snr)
        oa = orc.assemble(sub, sn, snr)
        ov = orc.msa(sub, sn, snr, oa)
        orc.genotype(sub, sn, snr, oa, ov, debug=False)
        ct = time.perf_counter() - c0
        cpu = {"value": round(sn / ct, 3), "unit": "windows/s", "cores": 1, "kind": "port",
               "sample": f"{sn} windows of the same {args.config} workload through the whole path (oracle, 1 thread, {ct:.1f} s)"}
        del og

    if rank == 0:
        out = {
            "metric": "microassembly windows/sec (whole node)", "value": round(wps, 2), "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/i32 (f64 statistics)", "data": "synthetic",
            "config": {"workload": f"{args.config}: chr22-shaped tumour/normal 30x/30x, 1001 bp windows, 150 bp paired reads, k=25 "
                                   f"(BASELINE.json configs[1])" if args.config == "C2" else args.config,
                       "windows_per_step_per_gpu": n, "distinct_windows": n0, "reads_per_window": round(R, 1),
                       "assembled_windows_per_s": round(asm_wps, 2), "assembled_fraction": round(assembled / n, 4),
                       "windows_with_capacity_overflow": overflowed, "haplotypes_per_assembled_window": round(H, 2),
                       "sharding": "static, one process per GPU, no collective"},
            "roofline": roof, "cpu_baseline": cpu, "kernel_ms_per_step": kernel_ms_per_step,
        }
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
