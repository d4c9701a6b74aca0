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
WORKLOADS = {
    "C3": "C3: whole-genome-shaped tumour/normal 60x/30x, 1001 bp windows, 150 bp paired reads, k=25 (the workload "
          "BASELINE.json's metric is quoted on = configs[2]; every GPU runs its own shard of windows)",
    "C2": "C2: chr22-shaped tumour/normal 30x/30x, 1001 bp windows, 150 bp paired reads, k=25 (BASELINE.json configs[1])",
    "C4": "C4: deep panel 500x/500x with 50 bp indels (BASELINE.json configs[3])",
    "C5": "C5: 1 tumour + 2 normals, 30x each (BASELINE.json configs[4])",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--windows", type=int, default=8192, help="windows per step per GPU")
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic windows (tiled to --windows)")
    ap.add_argument("--config", default="C3",
                    help="C3 = WGS-shaped tumour/normal 60x/30x (the workload BASELINE.json's metric is quoted on); "
                         "C2 = chr22-shaped 30x/30x (configs[1]); C4, C5")
    ap.add_argument("--no-also", action="store_true", help="skip the short secondary measurement of the other WGS config")
    ap.add_argument("--cpu-windows", type=int, default=64, help="oracle sample for cpu_baseline (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    return ap.parse_args()


def algorithmic_bytes(kernel, st):
    """Algorithmic HBM bytes per STEP of `kernel`: inputs read once, outputs written once, intermediates
    counted only where the design keeps them in HBM (DESIGN.md "Kernels" states each term).  Per-window
    figures x windows per step; the caller divides by the kernel's launches per step."""
    n, S = st["windows"], st["S"]
    W, R, Bb, Ni, Nn = st["W"], st["R"], st["B"], st["N_inst"], st["N_nodes"]
    Nslow, Ngen = st["N_slow"], st["N_gen"]
    H, L, m, band = st["H"], st["L"], st["read_len"], st["band"]
    P, Pdp = st["pairs_per_window"], st["dp_pairs_per_window"]
    Wk = max(W - st["k"] + 1, 0)
    tb_words = (2 * band + 1 + 7) // 8
    per = {
        "gate_kernel": W + 8,
        "k_count_inst": 12 * R,
        # bases + quals + read meta in, one instance word per k-mer out, slow queue out
        "k_classify": 2 * Bb + 12 * R + W + 4 * Ni + 4 * Nslow,
        # reference + slow k-mers hashed from their bytes, 16 B table probe/claim + first-instance update each
        "k_insert": W + st["k"] * Nslow + 20 * (Nslow + Wk) + 8 * Nslow,
        "k_support": 4 * Ni + 4 * (S + 2) * Wk,
        "k_mm_insert": 4 * Ni + 12 * Ngen,
        "k_count": 4 * Ni + 12 * Ngen + 8 * Ngen,
        # table scan + instance words + node records / edge slots out
        "k_rank": (16 + 4 * (S + 2)) * st["table_slots"] + 4 * Ni + (24 + 4 * S + 128) * Nn,
        "k_edges": 4 * Ni + 16 * (Nslow + Wk),
        "k_edge_sort": 128 * Nn,
        # node records + edge lists in, haplotype bases / runs / stats out
        "k_clean": (24 + 4 * S) * Nn + 64 * Nn + H * L + 64 * H + 256,
        # k_msa (graph update, traceback, variants): haplotypes in; per alignment the window's LDS image (graph +
        # state, st["poa_img"] bytes) saved and restored once and the decision codes read back along the path
        "k_msa": H * L + max(H - 1, 0) * (2 * st.get("poa_img", 78000) + 2 * 2 * L) + 512,
        # k_msa_band (the DP fill): row descriptors (10 B/row) + haplotype in; one 2-byte decision code per cell of
        # the 256-column band and H(i, L) per row out (HBM-resident by design: the codes do not fit LDS)
        "k_msa_band": max(H - 1, 0) * (10 * (L + 1) + L + 2 * 256 * (L + 1) + 4 * (L + 1)),
        "k_plan": 16,
        # every pair: read bases + 32 B result or 8 B list entry; haplotype once per (window, haplotype)
        "k_vote": P * (m + 32) + H * L,
        # DP pairs: read + haplotype segment in, 4-bit move codes of the band out and the path read back, result out
        "k_align_reg": Pdp * (m + (m + 2 * band + 1) + (m + 1) * tb_words * 4 + 4 * (m + 1) + 32),
        "k_align": Pdp * (m + (m + 2 * band + 1) + (m + 1) * tb_words * 4 + 4 * (m + 1) + 32),
        "k_assign": R * (m + H * 40 + 16),
        "k_evidence": R * 80,
        "k_qual": 64,
        # annotation: REF haplotype of each component in, one f64 out; +-50 windows of the REF and of
        # every ALT site in, 11 features + 3 graph metrics (88 B) out per variant
        "k_hap_lq": L + 8,
        "k_seqcx": st.get("V", 0) * (3 * 110 + 88),
    }
    return per.get(kernel, 0) * n


_START = None  # multi-worker baseline: every worker finishes synthesising its windows before any of them is timed


def _oracle_chunk(job):
    """One worker of the CPU baseline: the whole path (oracle) over `count` windows starting at `first`."""
    config, first, count, num_samples = job
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from harness import OracleEngine
    from lancet2_amd import capi, synth
    params = capi.default_params(min_k=25, max_k=25, num_samples=num_samples)
    sub, sn, snr = synth.make_config_batch(config, count, first_index=first)
    orc = OracleEngine(params)
    if _START is not None:
        try:
            _START.wait(timeout=120)
        except Exception:
            pass
    t0 = time.perf_counter()
    orc.gate(sub, sn, snr)
    oa = orc.assemble(sub, sn, snr)
    ov = orc.msa(sub, sn, snr, oa)
    orc.genotype(sub, sn, snr, oa, ov, debug=False)
    orc.annotate(sub, sn, snr, oa, ov)
    return sn, time.perf_counter() - t0


def cpu_baselines(args, num_samples):
    """Oracle (a port of the reference path) on a bounded sample of the same workload: (i) one thread -- the
    reported cpu_baseline -- and (ii) one worker process per host core, the reference's own threading model
    (pipeline_executor.cpp:174-197).  Runs BEFORE the GPU is initialised (it forks)."""
    import multiprocessing as mp
    n1 = args.cpu_windows
    sn, ct = _oracle_chunk((args.config, 10_000, n1, num_samples))
    cpu = {"value": round(sn / ct, 3), "unit": "windows/s", "cores": 1, "kind": "port",
           "sample": f"{sn} windows of the same {args.config} workload through the whole path (oracle, 1 thread, {ct:.1f} s)"}
    cores = os.cpu_count() or 1
    per = 4  # windows per worker: the leg stays around half a minute even when the box schedules far fewer cores than it reports
    jobs = [(args.config, 10_000 + 1000 * i, per, num_samples) for i in range(cores)]
    global _START
    ctx = mp.get_context("fork")
    _START = ctx.Barrier(cores)  # inherited by the forked workers
    out_q = ctx.Queue()

    def _worker(job):
        out_q.put(_oracle_chunk(job))

    procs = [ctx.Process(target=_worker, args=(j,)) for j in jobs]  # exactly one process per job (the barrier counts them)
    for pr in procs:
        pr.start()
    res = [out_q.get(timeout=900) for _ in procs]
    for pr in procs:
        pr.join()
    _START = None
    tot = sum(r[0] for r in res)
    slowest = max(r[1] for r in res)  # workers run concurrently; input synthesis (Python) is not timed
    cpu_mt = {"value": round(tot / slowest, 3), "unit": "windows/s", "cores": cores, "kind": "port",
              "sample": f"{tot} windows, one oracle process per host core ({cores} x {per} windows), slowest worker {slowest:.1f} s"}
    return cpu, cpu_mt


def main():
    args = parse()
    world0 = int(os.environ.get("WORLD_SIZE", "1"))
    cpu = cpu_mt = None
    if world0 == 1 and not args.no_cpu and args.cpu_windows > 0:
        cpu, cpu_mt = cpu_baselines(args, 3 if args.config == "C5" else 2)
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
    cx = dev_alloc(capi.cx_out_spec(params, n))
    gs, as_, vs, qs = (capi.fill_struct(capi.GateOut, g), capi.fill_struct(capi.AsmOut, a),
                       capi.fill_struct(capi.VarOut, v), capi.fill_struct(capi.GenoOut, q))
    cxs = capi.fill_struct(capi.CxOut, cx)
    eng = Engine(params, device=local_rank, memspace=capi.MA_MEM_DEVICE)
    stream = torch.cuda.current_stream(dev)
    eng.set_stream(stream.cuda_stream)

    def step():  # the metric's path: gate -> assembly -> POA/variants -> genotyping
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
    stats = eng.stats()
    # SEQ_CX / GRAPH_CX annotation of the batch's variants (a "next" row: not part of the metric's
    # path) -- timed on its own, after the timed region, and reported beside it
    eng.timing_control(0)
    eng.annotate_device(b, as_, vs, cxs, 0.41)
    barrier()
    eng.timing_control(2)
    ta = time.perf_counter()
    for _ in range(args.steps):
        eng.annotate_device(b, as_, vs, cxs, 0.41)
    barrier()
    annot_ms = (time.perf_counter() - ta) * 1e3 / max(args.steps, 1)
    annot_k = {}
    for name, ms in eng.kernel_times():
        annot_k[name] = annot_k.get(name, 0.0) + ms / max(args.steps, 1)
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
    steps = max(args.steps, 1)
    pairs_w = stats.get("pairs", 0) / steps / n
    dp_w = stats.get("dp_pairs", 0) / steps / n
    read_len = Bb / max(R, 1)
    # slow (non reference-identical) instances and general mate-mer instances: measured shares of the C2 workload
    # (profiles/r1_*: 12 % of the read k-mers take the hash-table path, 10 % go through the general mate-mer set)
    st = dict(windows=n, S=params.num_samples, W=W, R=R, B=Bb, N_inst=Ni, N_slow=0.12 * Ni, N_gen=0.10 * Ni,
              N_nodes=1.4 * W, table_slots=8192, H=H, L=L, read_len=read_len, band=params.band, k=k,
              pairs_per_window=pairs_w, dp_pairs_per_window=dp_w, V=float(nvars.sum()) / n)

    # ---- short secondary measurement of the other WGS-shaped config (reported beside the metric, never as `value`) ----
    also = None
    other = {"C3": "C2", "C2": "C3"}.get(args.config)
    if other and world == 1 and not args.no_also:
        o_arrs, o_n0, o_nr0 = synth.make_config_batch(other, distinct, first_index=10_000)
        o_arrs, o_n, o_nr = synth.tile_batch(o_arrs, o_n0, o_nr0, max(1, args.windows // o_n0))
        o_dbatch = {k: torch.from_numpy(v_.view(np.uint8) if v_.dtype != np.uint8 else v_).to(dev) for k, v_ in o_arrs.items()}
        o_b = capi.make_batch_struct(o_dbatch, o_n, o_nr)
        o_q = dev_alloc(capi.geno_out_spec(params, o_n, o_nr, debug=False))
        o_qs = capi.fill_struct(capi.GenoOut, o_q)
        eng.timing_control(0)
        eng.process_device(o_b, gs, as_, vs, o_qs)
        barrier()
        t_o = time.perf_counter()
        for _ in range(2):
            eng.process_device(o_b, gs, as_, vs, o_qs)
        barrier()
        also = {"workload": WORKLOADS[other], "value": round(2 * o_n / (time.perf_counter() - t_o), 2), "unit": "windows/s",
                "steps": 2, "windows_per_step": o_n}

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
        traffic = None
        tpath = os.path.join(REPO, "profiles", "r1_hbm_traffic.json")  # PMC pass (tools/hbm_traffic.py), bytes per launch
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom, {}).get("bytes_per_launch")
            except Exception:
                traffic = None
        ach = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": traffic,
                "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches_per_step,
                "algorithmic_bytes_per_launch": int(bytes_per_launch)}
    kernel_ms_per_step = {kname: round(val[0] / args.steps, 3) for kname, val in sorted(agg.items(), key=lambda kv: -kv[1][0])}

    # ---- per-stage table: HIP-event time, algorithmic bytes, achieved GB/s; cell rates of the two DP kernels ----
    stages = {}
    for kname, (tot_ms, launches) in agg.items():
        ab = algorithmic_bytes(kname, st)
        ms_step = tot_ms / args.steps
        if ab <= 0 or ms_step <= 0:
            continue
        gbs = ab / (ms_step * 1e-3) / 1e9
        stages[kname] = {"ms_per_step": round(ms_step, 3), "algorithmic_MB_per_step": round(ab / 1e6, 1),
                         "GB_per_s": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 5)}
    cells = {}
    if "k_msa_band" in agg:
        c = n * max(H - 1, 0) * (L + 1) * 256  # every non-first haplotype: 256-column band over the (L+1)-row graph
        cells["k_msa_band_GCUPS"] = round(c / (agg["k_msa_band"][0] / args.steps * 1e-3) / 1e9, 1)
    if "k_align_reg" in agg:
        c = n * dp_w * read_len * (2 * params.band + 1)
        cells["k_align_reg_GCUPS"] = round(c / (agg["k_align_reg"][0] / args.steps * 1e-3) / 1e9, 1)

    if rank == 0:
        out = {
            "metric": "microassembly windows/sec (whole node)", "value": round(wps, 2), "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/i32 (f64 statistics)", "data": "synthetic",
            "config": {"workload": WORKLOADS.get(args.config, args.config),
                       "windows_per_step_per_gpu": n, "distinct_windows": n0, "reads_per_window": round(R, 1),
                       "assembled_windows_per_s": round(asm_wps, 2), "assembled_fraction": round(assembled / n, 4),
                       "windows_with_capacity_overflow": overflowed, "haplotypes_per_assembled_window": round(H, 2),
                       "sharding": "static, one process per GPU, no collective"},
            "roofline": roof, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_mt,
            "kernel_ms_per_step": kernel_ms_per_step, "stages": stages, "dp_cell_rates": cells,
            "work": {"pairs_per_window": round(pairs_w, 1), "dp_pairs_per_window": round(dp_w, 1)},
            # SURVEY 8 f3 (next row), outside the metric's timed region: ma_annotate_batch over the same batch
            "also": also,
            "annotation": {"ms_per_step": round(annot_ms, 3), "variants_per_window": round(float(nvars.sum()) / n, 2),
                           "kernel_ms_per_step": {k_: round(v_, 3) for k_, v_ in annot_k.items()}},
        }
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
