#!/usr/bin/env python3
"""bench.py -- microassembly windows/sec on MI355X (BASELINE.json metric).

A step = one pass of the whole hot path (repeat gate -> cbdg assembly -> POA/variants -> read<->haplotype
genotyping: ma_process_batch) over one batch of synthetic tumour/normal windows that is already resident
in HBM.  N > 1: one process per GPU (torch.distributed / RCCL only for the barrier + max-over-ranks),
windows statically sharded, NO data-path collective (windows never communicate) -> weak scaling.
`python bench.py --gpus N` without a launcher starts the N ranks itself (torch.distributed.run as a child
process, before this process has touched a GPU); under an external torchrun it is one of the ranks.

Prints ONE JSON line on rank 0.  Roofline bytes are SURVEY.md 8(d)'s algorithmic bytes and nothing else: every
stage reads its inputs once and writes its outputs once; traceback tiles, decision codes, LDS images and table
re-scans are intermediates and are NOT counted (DESIGN.md section 5 restates the terms).
"""
import argparse
import json
import os
import subprocess
import sys
import time

# ROCm gives a process four hardware queues unless told otherwise; the engine's concurrent lanes + the caller's stream want
# five (set before anything initialises HIP; a value already in the environment wins)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# 256 CUs x 4 SIMDs x 32 lanes/cycle x 2.4 GHz (same guide: a wave64 VALU instruction issues over 2 cycles)
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9  # = 78.6e12 lane-operations/s
WORKLOADS = {
    "C3": "C3: whole-genome-shaped tumour/normal 60x/30x, 1001 bp windows, 150 bp paired reads, k=25 (the workload "
          "BASELINE.json's metric is quoted on = configs[2]; every GPU runs its own shard of windows)",
    "C2": "C2: chr22-shaped tumour/normal 30x/30x, 1001 bp windows, 150 bp paired reads, k=25 (BASELINE.json configs[1])",
    "C4": "C4: deep panel 500x/500x with 50 bp indels (BASELINE.json configs[3])",
    "C5": "C5: 1 tumour + 2 normals, 30x each (BASELINE.json configs[4])",
}
STR_UNITS = (b"A", b"CA", b"CAG", b"GATA", b"TTTCC", b"AGGGTT")
# which kernels make up which SURVEY 8(a) stage
STAGE_OF = {"gate_kernel": "gate",
            "k_count_inst": "build", "k_classify": "build", "k_insert": "build", "k_support": "build",
            "k_mm_lds": "build", "k_mm_insert": "build", "k_count": "build", "k_rank": "build", "k_edges": "build",
            "k_edge_sort": "build", "k_graph": "build", "k_graph_gen": "build", "k_mm_q": "build", "k_mm_hbm": "build",
            "k_clean": "clean", "k_clean_chains": "clean", "k_clean_tail": "clean",
            "k_msa": "poa", "k_msa_band": "poa", "k_poa": "poa",
            "k_read_planes": "genotype", "k_plan": "genotype", "k_vote": "genotype", "k_dp_scatter": "genotype", "k_align_reg": "genotype", "k_align_tb": "genotype",
            "k_align_wave": "genotype", "k_align_gen": "genotype", "k_assign": "genotype", "k_evidence": "genotype",
            "k_qual": "genotype"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--windows", type=int, default=16384, help="windows per step per GPU (round 5: 16384 -- a whole-genome run holds "
                                                               "~3 M windows; rounds 1-4 stepped 8192 at a time)")
    ap.add_argument("--distinct", type=int, default=16384, help="distinct synthetic windows (tiled to --windows if fewer)")
    ap.add_argument("--str-every", type=int, default=8, help="every n-th window carries a short tandem repeat (0 = none)")
    ap.add_argument("--nohint-every", type=int, default=50,
                    help="every n-th read pair arrives without a mapping hint (unmapped / rescued mates); 0 = every read hinted")
    ap.add_argument("--hard-every", type=int, default=16,
                    help="every n-th window carries a tandem duplication in the sample (30-80 bases: a cycle at k = 25, the k "
                         "ladder), another a low-complexity stretch, every 2n-th a dispersed duplication of the reference "
                         "(the repeat gate); 0 = none")
    ap.add_argument("--softclip", type=float, default=0.03, help="fraction of reads with an unrelated low-quality head or tail")
    ap.add_argument("--nfrac", type=float, default=0.01, help="fraction of reads with one to three N bases")
    ap.add_argument("--config", default="C3",
                    help="C3 = WGS-shaped tumour/normal 60x/30x (the workload BASELINE.json's metric is quoted on); "
                         "C2 = chr22-shaped 30x/30x (configs[1]); C4, C5")
    ap.add_argument("--c4-windows", type=int, default=512, help="distinct deep-panel windows of the also.c4_panel leg (0 = skip)")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary measurements (k cascade, other config, host path)")
    ap.add_argument("--cpu-windows", type=int, default=64, help="oracle sample for cpu_baseline (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--parity-windows", type=int, default=16,
                    help="N > 1 only (at N = 1 the cpu_baseline legs' windows are the parity sample): rank 0 runs the oracle on this many "
                         "of ITS OWN windows before the GPU is initialised and checks the engine's outputs of the last timed step "
                         "against them (0 = skip)")
    ap.add_argument("--gen-workers", type=int, default=0, help="processes that synthesise windows (0 = min(16, cores))")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="rehearsal of the N > 1 launch path on a box with fewer GPUs than ranks: every rank runs on device "
                         "(rank mod GPUs present), each planning its workspaces for its share of the HBM, and the barrier / "
                         "max-over-ranks go through gloo (RCCL cannot put two ranks on one device).  The line says so "
                         "(`oversubscribed`): it exercises the launcher, the sharding and the rank plumbing -- it is NOT a scaling "
                         "measurement")
    ap.add_argument("--gen-only", action="store_true", help="synthesise the windows (into MA_BENCH_CACHE) and exit: no GPU is touched")
    return ap.parse_args()


# ---- algorithmic bytes per window-attempt, by stage ------------------------------------------------
def survey_bytes(stage, st):
    """Per WINDOW (per k attempt for build/clean) -- the caller multiplies by windows / attempts per step.
    W window bases, R reads, B read bases, S samples, N_inst k-mer instances, N_raw distinct k-mers, H haplotypes,
    L mean haplotype length, var_bases allele bases written by the POA stage, n_cigar = record stride (max_cigar)."""
    W, R, B, S = st["W"], st["R"], st["B"], st["S"]
    Ni, Nraw, H, L = st["N_inst"], st["N_raw"], st["H"], st["L"]
    if stage == "gate":
        return W + 4
    if stage == "build":
        return 2 * B + 5 * R + W + 16 * Ni + (16 + 4 * S) * Nraw
    if stage == "clean":
        return (16 + 4 * S) * Nraw + 5 * H * L + 64
    if stage == "poa":
        return 5 * H * L + 16 * (st["L_ref"] + st["var_bases"])
    if stage == "genotype":
        return 2 * B + H * L + (24 + 4 * st["n_cigar"]) * R * H
    return 0


def kernel_bytes(kname, st):
    """DESIGN.md section 4, column "alg. bytes / window": what ONE kernel has to read and write per window (per k attempt for the
    build / clean kernels, per assembled window behind the assembler) when each of its inputs is read once and each of its
    outputs written once.  Ns slow-path instances, Ne (k+1)-mers of reads queued for the edge builder, Nc read-support counts
    queued, Nn nodes after the first low-coverage pass, slots = table slots per window (k_insert's direct map), P pairs,
    Pdp pairs that run the DP, m read length.  None: no formula (helper kernels)."""
    W, R, B, S = st["W"], st["R"], st["B"], st["S"]
    Ni, Ns, Ne, Nc, Nn = st["N_inst"], st["N_slow"], st["N_edgeq"], st["N_cntq"], st["N_nodes"]
    H, L, m, P, Pdp = st["H"], st["L"], st["m"], st["P"], st["P_dp"]
    slots, CW = 6144, S + 2
    table = {
        "gate_kernel": W + 8,
        "k_classify": 2 * B + 12 * R + W + 4 * Ni + 4 * Ns,
        "k_insert": W + B + 4 * Ns + 2 * 16 * (Ns + W) + 4 * (Ns + W) + (12 + 4 * CW) * slots,
        "k_support": 4 * Ni + 13 * R + 8 * Ne + 4 * Nc + 4 * CW * W,
        "k_graph": (12 + 4 * CW) * slots + 4 * Nc + 8 * Ne + 8 * W + (4 * S + 8 + 4 + 4) * Nn + 8 * Nn,
        "k_clean_chains": (4 * S + 8 + 4 + 4 + 16) * Nn + 4000,
        "k_clean_tail": 4000 + H * L + 64 * H,
        "k_msa": H * L + max(H - 1, 0) * 2 * 77_000 + 16 * (L + st["var_bases"]),
        # the banded fills that actually RAN (statistics step: band cells and fills per assembled window): one code byte
        # written per band cell, a 14-byte row descriptor and the haplotype read per fill -- half of the alignments are
        # written down in closed form and never reach this kernel (rounds 3-5 billed every alignment a full 128-column band)
        "k_msa_band": st.get("poa_band_cells", 0.0) + st.get("poa_band_fills", 0.0) * (14 * (L + 1) + L),
        # the persistent kernel (round 6) = the graph phases + the fills in one launch: haplotypes and variant records as k_msa,
        # a code byte written and read back per band cell, per fill the 14-byte row descriptors and one save + restore of the
        # window's LDS image (77 KB) -- not one per window and round
        "k_poa": H * L + 16 * (L + st["var_bases"]) + 2 * st.get("poa_band_cells", 0.0) +
                 st.get("poa_band_fills", 0.0) * (14 * (L + 1) + L + 2 * 77_000),
        "k_read_planes": B + 12 * (m / 32 + 2) * R,
        "k_vote": P * (12 * (m / 32 + 2) + 4 + 32) + H * L,
        "k_align_reg": Pdp * (2 * m + 45 / 2 * (m + 1) + 68),
        "k_assign": R * (2 * m + H * (24 + 68)) + R * st["max_vars"],
    }
    return table.get(kname)


def units_per_step(stage, st):
    """how many times a stage's per-window figure is paid per step: k attempts for build/clean, assembled windows
    for the stages behind the assembler"""
    if stage == "gate":
        return st["windows"]
    if stage in ("build", "clean"):
        return st["attempts_per_step"]
    return st["assembled"]


def pipeline_extract_leg(genome_len=2_400_000, depths=(30, 60), seed=0x5EED, threads=(1, -1), cxx_flags=()):
    """builds examples/pipeline_driver.cpp with g++, writes a random genome + two coordinate-sorted SAM files (150-base paired
    reads, plain 150M alignments) and times the extract stage on them: the driver prints each stage's busy time"""
    import re
    import shutil
    import subprocess
    import tempfile
    if shutil.which("g++") is None:
        raise RuntimeError("no g++")
    d = tempfile.mkdtemp(prefix="ma_extract_")
    try:
        exe = os.path.join(d, "pipeline_driver")
        lib = os.path.join(REPO, "lancet2_amd")
        subprocess.check_call(["g++", "-std=c++17", "-O3", "-march=native", os.path.join(REPO, "examples", "pipeline_driver.cpp"), "-I", os.path.join(REPO, "include"),
                               "-L", lib, "-lmicroasm", f"-Wl,-rpath,{lib}", "-Wl,--allow-shlib-undefined", "-DLANCET2_AMD_WITH_ZLIB", "-lz",
                               "-lpthread", "-o", exe] + list(cxx_flags))
        rng = np.random.default_rng(seed)
        genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, genome_len)]
        with open(os.path.join(d, "ref.fa"), "w") as f:
            f.write(">chr1\n" + bytes(genome).decode() + "\n")
        nreads = 0
        for name, depth in zip(("normal", "tumor"), depths):
            npairs = genome_len * depth // 300
            starts = np.sort(rng.integers(0, genome_len - 550, npairs))
            qual = "I" * 150
            recs = []
            for i, s0 in enumerate(starts):
                s1 = int(s0) + 250 + int(rng.integers(0, 150))
                recs.append((int(s0), f"{name[0]}{i}", 0x63, s1))
                recs.append((s1, f"{name[0]}{i}", 0x93, int(s0)))
            recs.sort()
            with open(os.path.join(d, name + ".sam"), "w") as f:
                f.write("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:%d\n" % genome_len)
                for pos0, qn, flag, mate in recs:
                    f.write(f"{qn}\t{flag}\tchr1\t{pos0 + 1}\t60\t150M\t=\t{mate + 1}\t{mate - pos0}\t{bytes(genome[pos0:pos0 + 150]).decode()}\t{qual}\tMD:Z:150\n")
            nreads += len(recs)
        best = None
        for nthreads in [effective_cores() if t_ < 0 else t_ for t_ in threads]:  # one collector thread, then one per usable core
            t0 = time.perf_counter()
            r = subprocess.run([exe, "--reference", os.path.join(d, "ref.fa"), "--normal", os.path.join(d, "normal.sam"), "--tumor",
                                os.path.join(d, "tumor.sam"), "--no-active-region", "--extract-only"] +
                               (["--extract-threads", str(nthreads)] if nthreads else []), capture_output=True, text=True)
            wall = time.perf_counter() - t0
            res = _parse_extract(r, wall, nreads, genome_len, depths)
            if best is None:
                best = res
                best["one_thread_windows_per_s"] = res["value"]
            elif res["value"] > best["value"]:
                res["one_thread_windows_per_s"] = best["one_thread_windows_per_s"]
                best = res
        return best
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _parse_extract(r, wall, nreads, genome_len, depths):
    import re
    if True:
        m = re.search(r"extract ([0-9.]+) s busy \(([0-9.]+) windows/s tiled, ([0-9.]+) shipped/s\) with (\d+) collector thread\(s\), ([0-9.]+) cpu-s of collection, ([0-9.]+) s of ordered batching", r.stderr)
        if r.returncode != 0 or not m:
            raise RuntimeError("pipeline_driver --extract-only: " + r.stderr[-200:])
        nwin = int(re.search(r"pipeline_driver: (\d+) windows", r.stderr).group(1))
        return {"value": float(m.group(3)), "unit": "windows/s handed to the engine by the extract stage", "busy_s": float(m.group(1)),
                "collector_threads": int(m.group(4)), "windows_per_cpu_second": round(nwin / max(float(m.group(5)), 1e-9), 1),
                "ordered_batching_s": float(m.group(6)),
                "windows": nwin, "reads": nreads, "reads_per_window": round(nreads * 1.25 / max(nwin, 1), 1),
                "wall_s_incl_sam_parsing": round(wall, 2),
                "note": "examples/pipeline_driver.cpp --extract-only --no-active-region on a random %d kb genome, %dx/%dx, SAM text; "
                        "one collector thread, then one per usable core (the reference runs one collector per worker, "
                        "pipeline_executor.cpp:174-197); the stage's span = max(slowest collector, ordered batching thread); the engine "
                        "takes ~250 k submitted windows/s" % (genome_len // 1000, depths[1], depths[0])}


# ---- synthetic windows ---------------------------------------------------------------------------------------------
NOHINT_EVERY = 50  # set from --nohint-every before any window is made (module global: the pool workers are forked)
HARD_EVERY, SOFTCLIP, NFRAC = 16, 0.03, 0.01  # likewise (--hard-every, --softclip, --nfrac)
TANDEM = (30, 45, 60, 80)


def window_kwargs(config, idx, str_every):
    """generator arguments of window `idx`: the WGS shape of `config` + what this index mixes in"""
    from lancet2_amd import synth
    k2 = dict(synth.CONFIGS[config])
    if str_every and idx % str_every == str_every - 1:
        k2["str_unit"] = STR_UNITS[(idx // str_every) % len(STR_UNITS)]
    if HARD_EVERY:
        h = HARD_EVERY
        if idx % h == 3:
            k2["tandem_dup"] = TANDEM[(idx // h) % len(TANDEM)]
        if idx % h == 9:
            k2["low_complexity"] = (60, 100)[(idx // h) % 2]
        if idx % (2 * h) == 13:
            k2["dup_len"] = 200
    if SOFTCLIP:
        k2["softclip_frac"] = SOFTCLIP
    if NFRAC:
        k2["n_frac"] = NFRAC
    return k2


def _gen_chunk(job):
    config, first, count, str_every = job[:4]  # `first` may also be the list of the chunk's window indices (count ignored)
    from lancet2_amd import capi, synth
    over = job[4] if len(job) > 4 else {}
    wins = []
    for idx in (first if isinstance(first, (list, tuple)) else range(first, first + count)):
        k2 = window_kwargs(config, idx, str_every)
        k2.update(over)
        w = synth.make_window(idx, **k2)
        if NOHINT_EVERY:
            for r in w["reads"]:
                if r["qname"] % NOHINT_EVERY == NOHINT_EVERY - 1:
                    r["hint"] = capi.MA_NO_HINT
        wins.append(w)
    return synth.pack_batch(wins)


def concat_batches(parts):
    """[(arrs, n, nr)] -> one packed batch"""
    out = {}
    n = sum(p[1] for p in parts)
    nr = sum(p[2] for p in parts)
    for k in ("ref_bases", "read_bases", "read_quals"):
        out[k] = np.concatenate([p[0][k][:-64] for p in parts] + [np.zeros(64, np.uint8)])

    def cat_off(key, dtype):
        chunks, base = [], np.uint64(0)
        for p in parts:
            off = p[0][key].astype(np.uint64)
            chunks.append(off[:-1] + base)
            base = base + off[-1]
        chunks.append(np.array([base], np.uint64))
        return np.concatenate(chunks).astype(dtype)

    out["ref_off"] = cat_off("ref_off", np.uint32)
    out["read_win_off"] = cat_off("read_win_off", np.uint32)
    out["read_off"] = cat_off("read_off", np.uint64)
    for k in ("read_qname_id", "read_sample", "read_flags", "read_hint"):
        out[k] = np.concatenate([p[0][k] for p in parts])
    return out, n, nr


def make_windows(config, count, first, str_every, workers, indices=None, over=None):
    """seeded windows [first, first + count) of `config` -- or the windows `indices` names, in that order -- synthesised by
    a pool of processes (forked BEFORE the GPU is initialised); the result does not depend on the number of workers"""
    import multiprocessing as mp
    per = 32
    over = over or {}
    # MA_BENCH_CACHE=<dir>: the synthesised batch is kept there (tools/r5_measure.sh: the profiler passes run the same command
    # seven times, each of which would otherwise synthesise its windows in ONE process -- forked pools do not survive rocprofv3)
    cache = os.environ.get("MA_BENCH_CACHE")
    cpath = None
    if cache:
        import hashlib
        key = repr((config, count, first, str_every, None if indices is None else list(indices), sorted(over.items()), NOHINT_EVERY,
                    HARD_EVERY, SOFTCLIP, NFRAC))
        cpath = os.path.join(cache, "batch_" + hashlib.sha256(key.encode()).hexdigest()[:16] + ".npz")
        if os.path.exists(cpath):
            z = np.load(cpath)
            return {k_: z[k_] for k_ in z.files if k_ not in ("_n", "_nr")}, int(z["_n"]), int(z["_nr"])
    if indices is None:
        jobs = [(config, first + o, min(per, count - o), str_every, over) for o in range(0, count, per)]
    else:
        jobs = [(config, list(indices[o:o + per]), 0, str_every, over) for o in range(0, len(indices), per)]
    workers = max(1, min(workers, len(jobs)))
    if workers == 1:
        parts = [_gen_chunk(j) for j in jobs]
    else:
        with mp.get_context("fork").Pool(workers) as pool:
            parts = pool.map(_gen_chunk, jobs, chunksize=1)
    res = concat_batches(parts)
    if cpath:
        os.makedirs(cache, exist_ok=True)
        np.savez(cpath, _n=res[1], _nr=res[2], **res[0])
    return res


_START = None  # multi-worker baseline: every worker finishes synthesising its windows before any of them is timed


def _oracle_chunk(job, keep=True):
    """One worker of the CPU baseline: the whole path (oracle) over `count` windows starting at `first` (or over the windows
    a list names).  keep: the oracle's outputs come back too -- they are what `parity_sample` checks the engine's outputs
    for the very same windows against after the timed region (the checker role; never the thing measured)."""
    config, first, count, num_samples, str_every = job
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from harness import OracleEngine
    from lancet2_amd import capi
    params = capi.default_params(min_k=25, max_k=25, num_samples=num_samples)
    sub, sn, snr = _gen_chunk((config, first, count, str_every))
    orc = OracleEngine(params)
    if _START is not None:
        try:
            _START.wait(timeout=120)
        except Exception:
            pass
    t0 = time.perf_counter()
    og = orc.gate(sub, sn, snr)
    oa = orc.assemble(sub, sn, snr)
    ov = orc.msa(sub, sn, snr, oa)
    oq = orc.genotype(sub, sn, snr, oa, ov, debug=False)
    dt = time.perf_counter() - t0
    kept = None
    if keep:
        kept = {"config": config, "windows": list(first) if isinstance(first, (list, tuple)) else list(range(first, first + sn)),
                "gate": og, "asm": oa, "var": ov, "geno": oq}
    return sn, dt, kept


def effective_cores():
    """cores this process can actually run on: its affinity mask, cut down to the cgroup's CPU quota (a box may list 256 CPUs
    and schedule 16 cores' worth of them)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], int(txt[1])
            else:
                quota, period = txt[0], int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and int(quota) > 0:
                n = min(n, max(1, int(quota) // period))
            break
        except Exception:
            continue
    return max(1, n)


def build_native_oracle():
    """the CPU baseline's build of the oracle: -O3 -march=native on THIS box; the parity tests keep
    the portable build.  Returns the library's path or None (then the portable build is timed and the line says so)."""
    try:
        subprocess.check_call(["make", "-B", "-s", "-C", os.path.join(REPO, "oracle"), "native"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL, timeout=300)
        path = os.path.join(REPO, "oracle", "liboracle_native.so")
        return path if os.path.exists(path) else None
    except Exception:
        return None


def cpu_baselines(args, num_samples):
    """Oracle (a port of the reference path) on a bounded sample of the same workload: (i) one thread -- the
    reported cpu_baseline -- and (ii) one worker process per core this process may use, the reference's own threading
    model (pipeline_executor.cpp:174-197).  Runs BEFORE the GPU is initialised (it forks)."""
    import multiprocessing as mp
    native = build_native_oracle()
    if native:
        os.environ["MA_ORACLE_LIB"] = native
    flags = "-O3 -march=native" if native else "-O2 (portable build: the native one did not build)"
    n1 = args.cpu_windows
    sn, ct, kept1 = _oracle_chunk((args.config, 10_000, n1, num_samples, args.str_every))
    kept = [kept1]
    cpu = {"value": round(sn / ct, 3), "unit": "windows/s", "cores": 1, "kind": "port",
           "sample": f"the first {sn} windows of the same {args.config} workload through the metric's path (gate, assembly, "
                     f"POA/variants, genotyping; oracle built {flags}, 1 thread, {ct:.1f} s)"}
    cores = effective_cores()
    per = 32  # windows per worker
    if ct / max(sn, 1) * per > 40.0:  # keep the leg around half a minute
        per = max(4, int(40.0 / (ct / max(sn, 1))))
    jobs = [(args.config, 10_000 + per * i, per, num_samples, args.str_every) for i in range(cores)]
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
    kept += [r[2] for r in res]
    cpu_mt = {"value": round(tot / slowest, 3), "unit": "windows/s", "cores": cores, "kind": "port",
              "sample": f"{tot} windows, one oracle process per usable core ({cores} x {per} windows; affinity mask cut to "
                        f"the cgroup quota, os.cpu_count() = {os.cpu_count()}), slowest worker {slowest:.1f} s"}
    # (iii) BASELINE.json configs[3]: a few deep-panel windows (7 k reads each; incl. windows whose LDS mate-mer set fills up
    #       and windows that end at the reference's traversal cap), one oracle process per window, each on one thread
    cpu_c4 = None
    if args.c4_windows > 0 and args.config == "C3" and not args.no_also:
        c4_idx = [i for i in C4_PARITY_WINDOWS if i < 10_000 + args.c4_windows]
        if c4_idx:
            jobs = [("C4", [i], 0, 2, 0) for i in c4_idx]
            with ctx.Pool(min(cores, len(jobs))) as pool:
                res4 = pool.map(_oracle_chunk, jobs, chunksize=1)
            kept += [r[2] for r in res4]
            t4 = sum(r[1] for r in res4)
            cpu_c4 = {"value": round(len(res4) / t4, 4), "unit": "windows/s", "cores": 1, "kind": "port",
                      "sample": f"{len(res4)} deep-panel windows of the c4_panel leg (indices {c4_idx}), one oracle process per "
                                f"window run side by side, value = windows / sum of the per-window times ({t4:.1f} s of CPU)"}
    os.environ.pop("MA_ORACLE_LIB", None)
    return cpu, cpu_mt, cpu_c4, kept


# deep-panel windows the parity sample covers (indices of the c4_panel leg's seeded windows): the two whose LDS mate-mer set
# fills up (the retry pass re-assembles them through the HBM set), windows that end at the reference's 2^20-pop traversal cap,
# windows over the caller's default haplotype caps, ordinary ones
C4_PARITY_WINDOWS = [10_488,                            # LDS mate-mer set fills up in the first pass (tools/dbg/c4_flags.py)
                     10_007, 10_011, 10_012, 10_016,    # end at the traversal cap
                     10_056, 10_098,                    # traversal cap AND more haplotypes than the default cap of 16
                     10_000, 10_002, 10_022]            # ordinary deep windows


def parity_check(kept, config, first, lp, n, out_dev, row_of=None):
    """The engine's outputs for the windows the cpu_baseline legs ran the oracle on, against the oracle's (gate, assembly,
    variants, allele counts / PL / GQ bit for bit, QUAL within 1e-9).  out_dev = (gate, asm, var, geno) dicts of device
    byte tensors of a batch of `n` windows whose window j is the seeded window first + j of `config`.
    -> (windows checked, mismatch strings)"""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from harness import compare_asm, compare_geno_calls, compare_vars
    from lancet2_amd import capi
    specs = (capi.gate_out_spec(n), capi.asm_out_spec(lp, n), capi.var_out_spec(lp, n), capi.geno_out_spec(lp, n, 0, debug=False))

    def rows(dev, spec, w0, cnt):
        got = {}
        for key, (dt, sz) in spec.items():
            per = sz // n * np.dtype(dt).itemsize
            got[key] = dev[key][w0 * per:(w0 + cnt) * per].cpu().numpy().view(dt)
        return got

    checked, bad = 0, []
    n_flagged = [0]
    for kp in kept:
        if kp is None or kp["config"] != config:
            continue
        wins = kp["windows"]
        if row_of is None:  # batch row of seeded window w (a sharded rank holds every world-th window: main() passes its map)
            runs = [(wins[0] - first, len(wins))] if wins == list(range(wins[0], wins[0] + len(wins))) else [(w - first, 1) for w in wins]
        else:
            runs = [(row_of(w), 1) for w in wins]
        if any(w0 < 0 or w0 + cnt > n for w0, cnt in runs):
            continue
        cnt_all = len(wins)
        got = [{k_: np.concatenate([rows(dev, spec, w0, cnt)[k_] for w0, cnt in runs]) for k_ in spec} for dev, spec in zip(out_dev, specs)]
        # A window that outgrew one of the CALLER's output caps (more haplotypes / longer haplotypes / more variants than the
        # fixed strides hold) is flagged by both sides and re-submitted by the host with larger buffers (include/microasm.h):
        # what its truncated rows hold is not part of the contract -- only that both sides flag it.
        over = np.uint32(capi.MA_W_HAP_OVERFLOW | capi.MA_W_LEN_OVERFLOW | capi.MA_W_VAR_OVERFLOW)
        flagged = (kp["asm"]["win_status"] & over) != 0
        if flagged.any():
            for gd, wd, spec in zip(got[1:], (kp["asm"], kp["var"], kp["geno"]), specs[1:]):
                for key in spec:
                    if key != "win_status":
                        gd[key].reshape(cnt_all, -1)[flagged] = wd[key].reshape(cnt_all, -1)[flagged]
            n_flagged[0] += int(flagged.sum())
        here = [] if np.array_equal(got[0]["max_approx"], kp["gate"]["max_approx"]) and \
            np.array_equal(got[0]["max_exact"], kp["gate"]["max_exact"]) else ["repeat gate differs"]
        here += compare_asm(lp, got[1], kp["asm"], cnt_all) + compare_vars(lp, got[2], kp["var"], cnt_all)
        here += compare_geno_calls(got[3], kp["geno"])
        bad += [f"{config} windows {wins[0]}..{wins[-1]}: {m}" for m in here[:4]]
        checked += cnt_all
    return checked, bad, n_flagged[0]


def count_gpus_sysfs():
    """GPUs of this node from the KFD topology (a node with simd_count > 0 is a GPU) -- no torch, no HIP: the process that
    starts the ranks must not have touched a GPU (a re-exec from a GPU-initialised process takes this pool's machines down).
    None when the topology is not readable (then the ranks find out for themselves)."""
    import glob
    if not os.path.exists("/dev/kfd"):
        return 0  # no compute driver at all
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    n = 0
    for path in nodes:
        try:
            for line in open(path):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
        except Exception:
            return None
    return n


def spawn_ranks(args):
    """--gpus N without a launcher: start the N ranks as children (one process per GPU) and relay rank 0's line.
    Nothing in THIS process touches a GPU or imports torch (tests/test_bench_launcher.py checks both)."""
    import socket
    have = count_gpus_sysfs()
    if have is not None and have < args.gpus and not getattr(args, "oversubscribe", False):
        print(json.dumps({"error": f"--gpus {args.gpus} but this node exposes {have} GPU(s)"}))
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    global NOHINT_EVERY, HARD_EVERY, SOFTCLIP, NFRAC
    NOHINT_EVERY = args.nohint_every
    HARD_EVERY, SOFTCLIP, NFRAC = args.hard_every, args.softclip, args.nfrac
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    num_samples = 3 if args.config == "C5" else 2
    workers = args.gen_workers or max(1, min(16, effective_cores() // max(1, world)))  # per rank: the ranks share the host
    # under rocprofv3 (--pmc initialises the GPU before main()) a forked pool never returns: synthesise in-process
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        workers = 1

    # ---- everything that forks happens before the GPU is initialised ----
    distinct = min(args.distinct, args.windows)
    # static sharding of ONE seeded window list over the ranks: block b of 32 windows -> rank b mod G (lancet2_amd/shard.py);
    # with one rank these are the windows 10 000 ... 10 000 + distinct - 1
    from lancet2_amd.shard import shard_indices
    first = 10_000
    # (dealt out in blocks of the workload's period: every 8th / 16th / 32nd window is of a difficult kind -- window i -> rank
    #  i mod G alone would hand rank 7 of 8 every tandem-repeat window and rank 0 none; 32 = the longest period)
    shard_block = 32 if distinct % 32 == 0 else 1
    mine = [first + i for i in shard_indices(distinct * world, rank, world, shard_block)]
    t_gen = time.perf_counter()
    arrs0, n0, nr0 = make_windows(args.config, distinct, first, args.str_every, workers, indices=mine)
    also_arrs = long_arrs = None
    other = {"C3": "C2", "C2": "C3"}.get(args.config)
    if other and world == 1 and not args.no_also:
        also_arrs = make_windows(other, min(distinct, 1024), first, args.str_every, workers)
        long_arrs = make_windows(args.config, min(distinct, 512), first, args.str_every, workers, over=dict(read_len=250))
    c4_arrs = c5_arrs = None
    if world == 1 and not args.no_also and args.config == "C3":
        # BASELINE.json configs[3] and [4]: the deep panel at full depth (500x/500x, 50 bp indels) and the three-sample mode
        c4_arrs = make_windows("C4", min(distinct, args.c4_windows), first, 0, workers) if args.c4_windows > 0 else None
        c5_arrs = make_windows("C5", min(distinct, 2048), first, args.str_every, workers)
    t_gen = time.perf_counter() - t_gen
    if args.gen_only:
        print(json.dumps({"synthesised_windows": n0, "s": round(t_gen, 1), "cache": os.environ.get("MA_BENCH_CACHE")}))
        return
    cpu = cpu_mt = cpu_c4 = None
    kept = []
    if world == 1 and not args.no_cpu and args.cpu_windows > 0:
        cpu, cpu_mt, cpu_c4, kept = cpu_baselines(args, num_samples)
    elif world > 1 and rank == 0 and args.parity_windows > 0:  # the checker on a few of rank 0's own windows (never timed)
        try:
            kept = [_oracle_chunk((args.config, mine[:min(args.parity_windows, len(mine))], 0, num_samples, args.str_every))[2]]
        except Exception as exc:  # (no oracle library on this box: the scaling run goes on without its parity sample, and says so)
            print(f"bench.py: parity sample skipped ({exc})", file=sys.stderr)
            kept = []

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        print(json.dumps({"error": "no GPU: the engine has no CPU fallback"}))
        sys.exit(2)
    oversub = bool(args.oversubscribe and world > 1)
    if oversub:  # (device_count() does not initialise the GPU on this image)
        local_rank = local_rank % max(1, torch.cuda.device_count())
        os.environ["MA_HBM_SHARE"] = str(round(0.9 / world, 3))  # the ranks share one device's HBM
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if oversub:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    from lancet2_amd import capi, synth
    from lancet2_amd.engine import Engine
    from lancet2_amd.stamp import build_stamp, csrc_sha16

    params = capi.default_params(min_k=25, max_k=25)  # BASELINE config: k = 25 (single attempt)
    params.num_samples = num_samples
    times = max(1, args.windows // n0)
    arrs, n, nr = synth.tile_batch(arrs0, n0, nr0, times)

    def to_dev(a_):
        return {k: torch.from_numpy(v_.view(np.uint8) if v_.dtype != np.uint8 else v_).to(dev) for k, v_ in a_.items()}

    dbatch = to_dev(arrs)
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

    # one untimed step in statistics mode: workload counters for the algorithmic-byte model
    eng.timing_control(3)
    step()
    torch.cuda.synchronize(dev)
    wstats = eng.stats()
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
    # ---- parity sample: the outputs of the LAST timed step for the windows the cpu_baseline legs ran the oracle on ----
    parity = None
    if kept:
        mine_row = {w_: j_ for j_, w_ in enumerate(mine)}
        row_of = None if world == 1 else (lambda w_: mine_row[w_])
        pc, pbad, pfl = parity_check(kept, args.config, first, params, n, (g, a, v, q), row_of=row_of)
        parity = {"windows": pc, "mismatches": len(pbad), "c4_windows": 0, "flagged_windows_status_only": pfl,
                  "checked": "repeat gate, haplotypes / weights / statistics (f64 bit patterns), variants and alleles, allele counts, "
                             "PL / GQ: equal; QUAL within 1e-9 -- engine outputs of the last timed step vs the oracle's for the "
                             "same windows (the cpu_baseline legs' own outputs); a window over one of the CALLER's output caps is flagged by both sides "
                             "and compared by its status word only (the host re-submits it with larger buffers)"}
        if pbad:
            parity["first_mismatches"] = pbad[:6]
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
    # ---- the measured device-copy ceiling SURVEY 8(d) asks for beside the 8 TB/s vendor figure: 1 GiB copied device to
    #      device (read + write = 2 GiB of HBM traffic per copy), after the timed region, torch events on torch's stream ----
    copy_ceiling = None
    try:
        cbytes = 1 << 30
        csrc_t = torch.empty(cbytes, dtype=torch.uint8, device=dev).fill_(7)
        cdst_t = torch.empty(cbytes, dtype=torch.uint8, device=dev)
        for _ in range(3):
            cdst_t.copy_(csrc_t)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            cdst_t.copy_(csrc_t)
        e1.record()
        torch.cuda.synchronize()
        copy_ceiling = round(2.0 * cbytes * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del csrc_t, cdst_t
    except Exception as ex:  # reported, never fatal: the ceiling is context for the roofline, not part of the metric
        copy_ceiling = None
        sys.stderr.write("bench: device-copy ceiling not measured: %r\n" % (ex,))
    rank_windows = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if oversub else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # which seeded windows each rank ran (first / last / count + a checksum of the indices): rank 0's line shows the shards
        mine_t = torch.tensor([mine[0], mine[-1], len(mine), sum(mine) % (1 << 31)], dtype=torch.int64, device="cpu" if oversub else dev)
        gathered = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(gathered, mine_t)
        rank_windows = [{"rank": r_, "first": int(g_[0]), "last": int(g_[1]), "count": int(g_[2]), "index_sum_mod_2_31": int(g_[3])}
                        for r_, g_ in enumerate(gathered)]

    # ---- workload statistics from the inputs, the results and the engine's counters ----
    status = a["win_status"].view(torch.int32).cpu().numpy().view(np.uint32)
    nvars = v["win_nvars"].view(torch.int32).cpu().numpy()
    hap_len = a["hap_len"].view(torch.int32).cpu().numpy().reshape(n, params.max_haps)
    nhaps = a["comp_nhaps"].view(torch.int32).cpu().numpy().reshape(n, params.max_comps)
    ncomp = a["win_ncomp"].view(torch.int32).cpu().numpy()
    gate_approx = g["max_approx"].view(torch.int32).cpu().numpy()
    alt_len = v["alt_len"].view(torch.int32).cpu().numpy().reshape(n, params.max_vars, params.max_alts)
    nalts = v["var_nalts"].view(torch.int32).cpu().numpy().reshape(n, params.max_vars)
    ref_len_v = v["var_ref_len"].view(torch.int32).cpu().numpy().reshape(n, params.max_vars)
    ok = (status & capi.MA_W_NO_HAPLOTYPE) == 0
    assembled = int(ok.sum())
    gated = int((gate_approx >= 25).sum())
    overflowed = int((status & (capi.MA_W_HAP_OVERFLOW | capi.MA_W_LEN_OVERFLOW | capi.MA_W_TABLE_OVERFLOW |
                                capi.MA_W_VAR_OVERFLOW)).astype(bool).sum())
    steps = max(args.steps, 1)
    attempts = max(wstats.get("window_attempts", 0), 1)   # of the one statistics step
    R = nr / n
    Bb = float(arrs["read_off"][-1]) / n
    W = float(arrs["ref_off"][-1]) / n
    cmask = np.arange(params.max_comps)[None, :] < ncomp[:, None]
    H = float(nhaps[cmask & ok[:, None]].sum()) / max(assembled, 1)
    hmask = np.arange(params.max_haps)[None, :] < np.where(cmask, nhaps, 0).sum(axis=1)[:, None]
    L = float(hap_len[hmask & ok[:, None]].sum()) / max(float((hmask & ok[:, None]).sum()), 1.0)
    vmask = (np.arange(params.max_vars)[None, :] < nvars[:, None]) & ok[:, None]
    amask = vmask[:, :, None] & (np.arange(params.max_alts)[None, None, :] < nalts[:, :, None])
    var_bases = float(ref_len_v[vmask].sum() + alt_len[amask].sum()) / max(assembled, 1)
    pairs_w = stats.get("pairs", 0) / steps / n
    dp_w = stats.get("dp_pairs", 0) / steps / n
    read_len = Bb / max(R, 1)
    st = dict(windows=n, S=params.num_samples, W=W, R=R, B=Bb,
              N_inst=wstats.get("kmer_instances", 0) / attempts, N_raw=wstats.get("distinct_kmers", 0) / attempts,
              H=H, L=L, L_ref=L, var_bases=var_bases, n_cigar=params.max_cigar,
              attempts_per_step=stats.get("window_attempts", 0) / steps, assembled=assembled)
    st.update(N_slow=wstats.get("slow_instances", 0) / attempts, N_edgeq=wstats.get("edge_queue", 0) / attempts,
              N_cntq=wstats.get("count_queue", 0) / attempts, N_nodes=wstats.get("nodes_after_lowcov", 0) / attempts,
              m=read_len, P=pairs_w * n / max(assembled, 1), P_dp=dp_w * n / max(assembled, 1), max_vars=params.max_vars)
    st.update(poa_band_cells=wstats.get("poa_band_cells", 0) / max(assembled, 1),
              poa_band_fills=wstats.get("poa_band_fills", 0) / max(assembled, 1))
    stage_bytes_step = {s: survey_bytes(s, st) * units_per_step(s, st) for s in ("gate", "build", "clean", "poa", "genotype")}

    # the metric counts ASSEMBLED windows: those that pass the repeat gate, yield haplotypes and run POA and
    # genotyping too -- counted on every rank and added up (the ranks' shards hold the same mix: blocks of the workload's
    # period; the sum does not depend on that)
    total_windows = n * args.steps * world
    wps_all = total_windows / elapsed
    assembled_all = assembled * world
    if world > 1:
        t_asm = torch.tensor([assembled], dtype=torch.int64, device="cpu" if oversub else dev)
        dist.all_reduce(t_asm, op=dist.ReduceOp.SUM)
        assembled_all = int(t_asm.item())
    asm_wps = assembled_all * args.steps / elapsed

    # ---- per-kernel / per-stage times ----
    agg = {}
    for name, ms in ktimes:
        s = agg.setdefault(name, [0.0, 0])
        s[0] += ms
        s[1] += 1
    kernel_ms_per_step = {kname: round(val[0] / args.steps, 3) for kname, val in sorted(agg.items(), key=lambda kv: -kv[1][0])}
    stage_ms = {}
    for kname, (tot_ms, _) in agg.items():
        sname = STAGE_OF.get(kname, "other")
        stage_ms[sname] = stage_ms.get(sname, 0.0) + tot_ms / args.steps
    stages = {}
    for s, ms in sorted(stage_ms.items(), key=lambda kv: -kv[1]):
        ab = stage_bytes_step.get(s, 0)
        gbs = ab / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        stages[s] = {"kernel_ms_per_step": round(ms, 3), "algorithmic_MB_per_step": round(ab / 1e6, 1),
                     "MB_per_window": round(survey_bytes(s, st) / 1e6, 4), "GB_per_s": round(gbs, 1),
                     "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 5),
                     "frac_of_copy_ceiling": round(gbs / copy_ceiling, 5) if copy_ceiling else None}
    step_bytes = sum(stage_bytes_step.values())

    # ---- dominant kernel: HBM roofline by its stage's SURVEY 8(d) bytes, and a VALU roofline from the PMC pass ----
    dom = max(agg.items(), key=lambda kv: kv[1][0])[0] if agg else None
    roof = roof_valu = None
    prof = {}
    def prof_is_stale(stamp):  # the counters were taken from another build than the one this run times
        # by the kernel SOURCES: two builds of the same sources differ as files (hipcc's fat binaries are not reproducible bit
        # for bit), so the library's own hash -- kept in the stamps for the record -- says nothing about a fresh build
        return stamp.get("csrc_sha16") != build_stamp()["csrc_sha16"]

    for cand in ("r6_pmc_per_kernel.json", "r5_pmc_per_kernel.json", "r4_pmc_per_kernel.json", "r3_pmc_per_kernel.json", "r2_pmc_per_kernel.json"):
        tpath = os.path.join(REPO, "profiles", cand)
        if os.path.exists(tpath):
            try:
                prof = json.load(open(tpath))
                prof["_file"] = "profiles/" + cand
                break
            except Exception:
                prof = {}
    if dom:
        tot_ms, launches = agg[dom]
        avg_ms = tot_ms / launches
        launches_per_step = launches / args.steps
        sname = STAGE_OF.get(dom, "other")
        own = kernel_bytes(dom, st)
        bytes_per_launch = (own * units_per_step(sname, st) if own is not None else stage_bytes_step.get(sname, 0)) / max(launches_per_step, 1)
        ach = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        pk = prof.get(dom) or next((v for k, v in prof.items() if k.startswith(dom) and isinstance(v, dict)), {})  # (k_align_reg -> k_align_reg2p)
        roof = {"bound": "hbm", "kernel": dom, "stage": sname, "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": pk.get("bytes_per_launch"),
                "traffic_source": prof.get("_file") if pk else None,
                # the PMC passes are a separate run of this command (rocprofv3 --pmc cannot share a run with the timing);
                # stale = the kernel sources have changed since those passes were taken
                "traffic_stale": prof_is_stale(prof.get("_stamp", {})) if pk else None,
                "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches_per_step,
                "algorithmic_bytes_per_launch": int(bytes_per_launch),
                # SURVEY 8(d) as written: the STAGE's algorithmic bytes per step / the summed time of the stage's kernels
                "frac_survey": stages.get(sname, {}).get("frac_of_hbm_peak"),
                # the same two against what a plain device-to-device copy of 1 GiB reaches on this GPU (read + write counted)
                "copy_ceiling_GBps": copy_ceiling,
                "frac_of_copy_ceiling": round(ach / copy_ceiling, 6) if copy_ceiling else None,
                "frac_survey_of_copy_ceiling": stages.get(sname, {}).get("frac_of_copy_ceiling"),
                "survey_stage_MB_per_step": stages.get(sname, {}).get("algorithmic_MB_per_step"),
                "survey_stage_kernel_ms_per_step": stages.get(sname, {}).get("kernel_ms_per_step"),
                "note": "achieved = this kernel's OWN algorithmic bytes per launch (bench.py: kernel_bytes = DESIGN.md section 4's per-window "
                        "figure x the units one launch processes) / its mean launch time (HIP events on the launch stream); "
                        "frac_survey = SURVEY 8(d)'s per-window bytes of the kernel's STAGE x the stage's units per step / the summed "
                        "time of the stage's kernels / peak (what `stages` lists for every stage); "
                        "roofline_top5 has the same for the five kernels with the most summed time"}
        if pk.get("valu_insts_per_launch"):
            lane_ops = pk["valu_insts_per_launch"] * 64.0
            av = lane_ops / (avg_ms * 1e-3)
            roof_valu = {"bound": "valu", "kernel": dom, "achieved": round(av / 1e12, 3), "peak": round(VALU_PEAK_LANE_OPS / 1e12, 1),
                         "unit": "T lane-ops/s", "frac": round(av / VALU_PEAK_LANE_OPS, 4),
                         "valu_insts_per_launch": pk["valu_insts_per_launch"], "source": prof.get("_file"),
                         "stale": prof_is_stale(prof.get("_stamp", {})),
                         "note": "SQ_INSTS_VALU (wave instructions, committed PMC pass of this command) x 64 lanes / this run's "
                                 "mean launch time; peak = 256 CU x 4 SIMD x 32 lanes x 2.4 GHz"}

    # ---- the five kernels with the most summed time, each priced with its OWN algorithmic bytes (kernel_bytes), beside what the
    #      committed PMC passes measured for it (traffic, vector instructions, share of wave cycles parked on a wait) ----
    sq = {}
    try:
        sq = json.load(open(os.path.join(REPO, "profiles", "r6_pmc_sq_per_kernel.json")))
    except Exception:
        sq = {}
    top5 = []
    for kname, (tot_ms, launches) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:5]:
        per_win = kernel_bytes(kname, st)
        sname = STAGE_OF.get(kname, "other")
        units = units_per_step(sname, st)
        lps = launches / args.steps
        avg_ms = tot_ms / launches
        ent = {"kernel": kname, "stage": sname, "avg_launch_ms": round(avg_ms, 4), "launches_per_step": lps,
               "ms_per_step_summed_over_lanes": round(tot_ms / args.steps, 3)}
        if per_win is not None:
            bpl = per_win * units / max(lps, 1)
            ent.update(algorithmic_bytes_per_window=int(per_win), algorithmic_bytes_per_launch=int(bpl),
                       achieved_GBps=round(bpl / (avg_ms * 1e-3) / 1e9, 2), frac_of_hbm_peak=round(bpl / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5))
        pk = prof.get(kname) or next((v_ for k_, v_ in prof.items() if k_.startswith(kname) and isinstance(v_, dict)), {})
        if pk:
            ent.update(traffic_bytes_per_launch=pk.get("bytes_per_launch"), valu_insts_per_launch=pk.get("valu_insts_per_launch"),
                       counters_stale=prof_is_stale(prof.get("_stamp", {})), counters_source=prof.get("_file"))
            if pk.get("valu_insts_per_launch"):
                ent["valu_frac"] = round(pk["valu_insts_per_launch"] * 64.0 / (avg_ms * 1e-3) / VALU_PEAK_LANE_OPS, 4)
            if pk.get("bytes_per_launch") and per_win is not None:
                ent["traffic_over_algorithmic"] = round(pk["bytes_per_launch"] / max(bpl, 1), 2)
        sk = sq.get(kname) or next((v_ for k_, v_ in sq.items() if k_.startswith(kname) and isinstance(v_, dict)), {})
        if sk.get("SQ_WAVE_CYCLES"):
            ent["wait_any_share"] = round(sk.get("SQ_WAIT_ANY", 0) / sk["SQ_WAVE_CYCLES"], 3)
        top5.append(ent)

    # ---- GCUPS of the two DP stages: cells of the statistics step / the summed time of the
    #      kernels that compute them in the timed region; beside it the vector lane-operations each cell cost (committed
    #      SQ_INSTS_VALU pass of this command) -- the VALU roofline is what bounds integer DP on this chip, not MFMA ----
    def pmc_insts_per_step(prefixes):
        """wave-level VALU instructions per step of every profiled kernel whose name starts with one of `prefixes`"""
        steps_prof = prof.get("_stamp", {}).get("steps_profiled", 5)  # 2 timed + 2 warm-up + 1 statistics step
        tot, found = 0.0, False
        for k_, v_ in prof.items():
            if isinstance(v_, dict) and not k_.startswith("_") and k_.startswith(tuple(prefixes)) and v_.get("valu_insts_per_launch"):
                tot += v_["valu_insts_per_launch"] * v_.get("launches", 0)
                found = True
        return tot / steps_prof if found else None

    def gcups_entry(cells, knames, what, extra_kernels=()):
        ms = sum(agg[k_][0] for k_ in knames if k_ in agg) / args.steps
        if cells <= 0 or ms <= 0:
            return None
        ent = {"cells_per_step": int(cells), "cells": what, "kernels": [k_ for k_ in knames if k_ in agg],
               "kernel_ms_per_step_summed_over_lanes": round(ms, 3), "GCUPS": round(cells / (ms * 1e-3) / 1e9, 1)}
        if extra_kernels:
            ms2 = ms + sum(agg[k_][0] for k_ in extra_kernels if k_ in agg) / args.steps
            ent["GCUPS_incl_" + "_".join(k_.replace("k_", "") for k_ in extra_kernels if k_ in agg)] = round(cells / (ms2 * 1e-3) / 1e9, 1)
        insts = pmc_insts_per_step(knames)
        if insts:
            ent.update(valu_lane_ops_per_cell=round(insts * 64.0 / cells, 1),
                       valu_frac_of_peak=round(insts * 64.0 / (ms * 1e-3) / VALU_PEAK_LANE_OPS, 4),
                       valu_peak_T_lane_ops_per_s=round(VALU_PEAK_LANE_OPS / 1e12, 1),
                       counters_source=prof.get("_file"), counters_stale=prof_is_stale(prof.get("_stamp", {})))
        return ent

    gcups = {
        "read_aligner": gcups_entry(wstats.get("aln_dp_cells", 0), ("k_align_reg", "k_align_wave", "k_align_gen"),
                                    "rows x region width of every read x haplotype pair that ran the DP (a19); the kernels compute "
                                    "their width class's cells in chunks of eight up to the widest region of a group", ("k_align_tb",)),
        "poa_band": gcups_entry(wstats.get("poa_band_cells", 0), ("k_msa_band", "k_poa"),
                                "rows x band columns of every banded haplotype <-> graph fill that ran (a18); alignments written "
                                "down in closed form have no cells"),
        "poa_alignments_per_step": wstats.get("poa_alignments", 0), "poa_closed_form_per_step": wstats.get("poa_direct_alignments", 0),
        "poa_band_fills_per_step": wstats.get("poa_band_fills", 0), "poa_full_fill_cells_per_step": wstats.get("poa_full_cells", 0),
        "dp_pairs_per_step": wstats.get("dp_pairs", 0),
        "note": "GCUPS = 1e9 cell updates per second of kernel time; under concurrent lanes the kernels' times are stretched by sharing "
                "the chip (profiles/*_bench_single_lane.json has the unshared times)"}

    # ---- secondary measurements (never `value`) ----
    also = {}
    eng.close()  # its workspaces are grow-only shares of the whole HBM: one engine at a time
    if world == 1 and not args.no_also:
        # (1) the default k cascade (min_k 13 .. max_k 127, graph_params.h:11-26) on the same windows
        cparams = capi.default_params()
        cparams.num_samples = num_samples
        ceng = Engine(cparams, device=local_rank, memspace=capi.MA_MEM_DEVICE)
        ceng.set_stream(stream.cuda_stream)
        ceng.timing_control(0)
        ceng.process_device(b, gs, as_, vs, qs)
        barrier()
        ceng.timing_control(2)
        t_c = time.perf_counter()
        for _ in range(2):
            ceng.process_device(b, gs, as_, vs, qs)
        barrier()
        dt = time.perf_counter() - t_c
        cst = ceng.stats()
        cstatus = a["win_status"].view(torch.int32).cpu().numpy().view(np.uint32)
        c_asm = float(((cstatus & capi.MA_W_NO_HAPLOTYPE) == 0).mean())
        also["k_cascade"] = {"workload": "the same windows with the reference's default cascade k = 13, 19, ... 127",
                             "value": round(2 * n * c_asm / dt, 2), "unit": "assembled windows/s", "steps": 2,
                             "submitted_windows_per_s": round(2 * n / dt, 2),
                             "k_attempts_per_window": round(cst.get("window_attempts", 0) / 2 / n, 2),
                             "assembled_fraction": round(c_asm, 4)}
        ceng.close()
        # (2) the other WGS-shaped config, and the headline's windows sequenced 2 x 250
        def device_leg(batch, label, **kw):
            # a secondary measurement must never cost the headline its line: an error becomes the leg's entry
            try:
                return device_leg_run(batch, label, **kw)
            except Exception as exc:
                return {"workload": label, "error": repr(exc)[:300]}

        def device_leg_run(batch, label, leg_params=None, max_windows=None, kernels=False, parity_config=None, resubmit=False, steps=2):
            lp = leg_params or params
            o_arrs, o_n0, o_nr0 = batch
            o_arrs, o_n, o_nr = synth.tile_batch(o_arrs, o_n0, o_nr0, max(1, min(args.windows, n, max_windows or args.windows) // o_n0))
            o_dbatch = to_dev(o_arrs)
            o_b = capi.make_batch_struct(o_dbatch, o_n, o_nr)
            o_q = dev_alloc(capi.geno_out_spec(lp, o_n, o_nr, debug=False))
            o_qs = capi.fill_struct(capi.GenoOut, o_q)
            oeng = Engine(lp, device=local_rank, memspace=capi.MA_MEM_DEVICE)
            oeng.set_stream(stream.cuda_stream)
            oeng.timing_control(0)
            oeng.process_device(o_b, gs, as_, vs, o_qs)
            barrier()
            # Windows that outgrew one of the CALLER's output caps (fixed strides: max_haps, max_hap_len, max_vars) are flagged;
            # the host re-submits them on a context with larger buffers, as examples/host_driver.cpp does (max_haps 32,
            # max_hap_len 4096, max_vars 256).  The flagged set of the warm-up step -- the steps run the same windows -- is
            # gathered into a batch of its own, and every timed step runs it right behind the main batch: the leg's time
            # includes the re-submission.
            redo = None
            if resubmit:
                wst = a["win_status"].view(torch.int32).cpu().numpy().view(np.uint32)[:o_n]
                capm = np.uint32(capi.MA_W_HAP_OVERFLOW | capi.MA_W_LEN_OVERFLOW | capi.MA_W_VAR_OVERFLOW)
                fl = np.nonzero((wst & capm) != 0)[0]
                if len(fl):
                    parts = [synth.slice_batch(o_arrs, o_n, int(w_), int(w_) + 1) for w_ in fl]
                    r_arrs, r_n, r_nr = concat_batches(parts)
                    big = capi.default_params(**{f_: getattr(lp, f_) for f_, _ in capi.Params._fields_})
                    big.max_haps, big.max_hap_len, big.max_vars, big.max_allele_bytes, big.max_runs = 32, 4096, 256, 16384, 512
                    r_db = to_dev(r_arrs)
                    r_out = [dev_alloc(capi.gate_out_spec(r_n)), dev_alloc(capi.asm_out_spec(big, r_n)),
                             dev_alloc(capi.var_out_spec(big, r_n)), dev_alloc(capi.geno_out_spec(big, r_n, r_nr, debug=False))]
                    r_structs = (capi.make_batch_struct(r_db, r_n, r_nr), capi.fill_struct(capi.GateOut, r_out[0]),
                                 capi.fill_struct(capi.AsmOut, r_out[1]), capi.fill_struct(capi.VarOut, r_out[2]),
                                 capi.fill_struct(capi.GenoOut, r_out[3]))
                    os.environ["MA_HBM_SHARE"] = "0.15"  # a second context beside the leg's: a small share of the HBM
                    reng = Engine(big, device=local_rank, memspace=capi.MA_MEM_DEVICE)
                    os.environ.pop("MA_HBM_SHARE", None)
                    # (on a stream of its own: the spare context works beside the main one -- the next batch's kernels run
                    #  while the few re-submitted windows go through their serial deep-window searches; barrier() waits for both)
                    r_stream = torch.cuda.Stream(dev)
                    reng.set_stream(r_stream.cuda_stream)
                    reng.timing_control(0)
                    reng.process_device(*r_structs)
                    barrier()
                    redo = (reng, r_structs, r_out, r_n, fl, r_db, r_stream)
            if kernels:
                oeng.timing_control(2)
            # the spare context has a host thread of its own (as examples/host_driver.cpp's has): the main context's next step
            # is submitted while the re-submitted windows of this one go through their serial deep-window searches
            redo_q = redo_th = None
            if redo:
                import queue
                import threading
                redo_q = queue.Queue()
                redo_err = []

                def redo_worker():
                    while redo_q.get() is not None:
                        try:
                            if not redo_err:
                                redo[0].process_device(*redo[1])
                        except Exception as exc:  # (raised again on the main thread below)
                            redo_err.append(exc)

                redo_th = threading.Thread(target=redo_worker)
                redo_th.start()
            t_o = time.perf_counter()
            for _ in range(steps):
                oeng.process_device(o_b, gs, as_, vs, o_qs)
                if redo:
                    redo_q.put(1)
            if redo:
                redo_q.put(None)
                redo_th.join()
                if redo_err:
                    raise redo_err[0]
            barrier()
            dt_o = time.perf_counter() - t_o
            ost = a["win_status"].view(torch.int32).cpu().numpy().view(np.uint32)[:o_n].copy()
            n_redone = 0
            if redo:  # the re-submitted windows' status words replace their first-pass ones
                rst = redo[2][1]["win_status"].view(torch.int32).cpu().numpy().view(np.uint32)[:redo[3]]
                ost[redo[4]] = rst
                n_redone = int(redo[3])
                redo[0].close()
            o_asm = float(((ost & capi.MA_W_NO_HAPLOTYPE) == 0).mean())
            res = {"workload": label, "value": round(steps * o_n * o_asm / dt_o, 2), "unit": "assembled windows/s", "steps": steps,
                   "submitted_windows_per_s": round(steps * o_n / dt_o, 2), "assembled_fraction": round(o_asm, 4),
                   "windows_per_step": o_n, "distinct_windows": o_n0, "reads_per_window": round(o_nr / o_n, 1)}
            if kernels:
                acc = {}
                for kname, ms in oeng.kernel_times():
                    acc[kname] = acc.get(kname, 0.0) + ms / steps
                res["kernel_ms_per_step"] = {k_: round(v_, 3) for k_, v_ in sorted(acc.items(), key=lambda kv: -kv[1])}
                flagged = ost & ~np.uint32(capi.MA_W_NO_HAPLOTYPE | capi.MA_W_BFS_LIMIT)
                res["windows_with_capacity_flag"] = int((flagged != 0).sum())
                res["capacity_flags"] = {nm: int(((ost & np.uint32(getattr(capi, nm))) != 0).sum())
                                         for nm in ("MA_W_HAP_OVERFLOW", "MA_W_LEN_OVERFLOW", "MA_W_TABLE_OVERFLOW", "MA_W_VAR_OVERFLOW",
                                                    "MA_W_CIGAR_OVERFLOW", "MA_W_READ_OVERFLOW")
                                         if ((ost & np.uint32(getattr(capi, nm))) != 0).any()}
                res["windows_at_traversal_limit"] = int(((ost & capi.MA_W_BFS_LIMIT) != 0).sum())
                if resubmit:
                    res["windows_resubmitted_with_larger_buffers"] = n_redone
            if parity_config and kept and parity is not None:  # the oracle ran some of this leg's windows too (cpu_baselines)
                pc, pbad, pfl = parity_check(kept, parity_config, first, lp, o_n, (g, a, v, o_q))
                res["parity_sample"] = {"windows": pc, "mismatches": len(pbad), "flagged_windows_status_only": pfl}
                parity["c4_windows"] += pc
                parity["flagged_windows_status_only"] += pfl
                parity["mismatches"] += len(pbad)
                if pbad:
                    parity.setdefault("first_mismatches", []).extend(pbad[:6])
            oeng.close()
            del o_dbatch, o_q
            return res

        if also_arrs is not None:
            also["other_config"] = device_leg(also_arrs, WORKLOADS[other])
        if long_arrs is not None:
            also["reads_2x250"] = device_leg(long_arrs, WORKLOADS[args.config] + " -- sequenced as 2 x 250 bp reads")
        if c4_arrs is not None:  # 7.1 k reads a window; a panel has ~10 k windows: 2048 per step (512 distinct, tiled)
            also["c4_panel"] = device_leg(c4_arrs, WORKLOADS["C4"], max_windows=max(c4_arrs[1], 2048), kernels=True, parity_config="C4",
                                          resubmit=True, steps=4)  # (four steps: the re-submitted windows of a step run under the next step)
            if cpu_c4:
                also["c4_panel"]["cpu_baseline"] = cpu_c4
            also["c4_panel_512"] = device_leg(c4_arrs, WORKLOADS["C4"] + " -- 512 windows per step, every one distinct", max_windows=c4_arrs[1])
        if c5_arrs is not None:
            p5 = capi.default_params(min_k=25, max_k=25)
            p5.num_samples = 3
            # the output structs of the headline (two samples) are large enough for the assembly / variant arrays; the
            # genotype arrays are allocated per leg from the leg's own parameters
            also["c5_three_samples"] = device_leg(c5_arrs, WORKLOADS["C5"], leg_params=p5, kernels=True)
        # (2b) the host shell's EXTRACT stage alone (examples/pipeline_driver.cpp --extract-only): window tiling,
        #      gates, read collection with the reference's filters / downsampling / comparator, Flatten -- how many windows
        #      per second ONE extract thread can hand the engine, on a synthetic 60x/30x genome read from SAM text
        try:
            also["pipeline_extract"] = pipeline_extract_leg()
        except Exception as exc:  # (no g++ / zlib on the box: not part of the metric)
            also["pipeline_extract"] = {"error": str(exc)[:200]}
        # (3) host path: caller-owned PINNED host buffers through MA_MEM_HOST -- the library stages inputs through HBM and
        #     copies every fixed-stride output array back (PCIe both ways inside the timed region).  One feeder = one
        #     context, nothing overlaps; two feeders = two contexts on the same device, each on its own host thread (what
        #     examples/host_driver.cpp --feeders 2 does): one batch's copies run under the other batch's kernels.
        try:
            import threading
            h_arrs, hn, h_nr = arrs, n, nr  # the headline's own batch, from the caller's (pinned) host memory
            keep = []

            def pinned(nbytes):
                t_ = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, pin_memory=True)
                keep.append(t_)
                return t_.numpy()

            def make_feeder():
                h_in = {}
                for k_, v_ in h_arrs.items():
                    v_ = np.ascontiguousarray(v_)
                    buf = pinned(v_.nbytes)[: v_.nbytes].view(v_.dtype)
                    buf[...] = v_
                    h_in[k_] = buf

                def pinned_out(spec):
                    return {k_: pinned(int(sz) * np.dtype(dt_).itemsize)[: int(sz) * np.dtype(dt_).itemsize].view(dt_)
                            for k_, (dt_, sz) in spec.items()}

                outs = (pinned_out(capi.gate_out_spec(hn)), pinned_out(capi.asm_out_spec(params, hn)),
                        pinned_out(capi.var_out_spec(params, hn)), pinned_out(capi.geno_out_spec(params, hn, h_nr, debug=False)))
                structs = (capi.fill_struct(capi.GateOut, outs[0]), capi.fill_struct(capi.AsmOut, outs[1]),
                           capi.fill_struct(capi.VarOut, outs[2]), capi.fill_struct(capi.GenoOut, outs[3]))
                return h_in, outs, structs, capi.make_batch_struct(h_in, hn, h_nr)

            def host_leg(n_feeders, steps_each=5, prefetch=True, drain=False):
                os.environ["MA_HBM_SHARE"] = str(round(0.9 / n_feeders, 3))
                feeders = [make_feeder() for _ in range(n_feeders)]
                engs = [Engine(params, device=local_rank, memspace=capi.MA_MEM_HOST) for _ in range(n_feeders)]
                for e_, f_ in zip(engs, feeders):
                    e_.timing_control(int(os.environ.get("MA_BENCH_HOST_TIMING", "2")))  # HIP events around every kernel, as in the timed region above
                    for _ in range(2 if prefetch else 1):  # warm-up: allocations, both input sets of the prefetching route
                        if prefetch:
                            e_.prefetch(f_[3])
                        e_.process_device(f_[3], *f_[2])
                start = threading.Barrier(n_feeders + 1)

                def work(e_, f_):
                    # steady state of a stream of batches: the pipeline is primed before the clock starts (as the warm-up
                    # steps of the headline are) and EVERY timed step uploads one batch and processes one -- the last
                    # step's upload is of a batch nobody processes, so that uploads and steps stay one to one
                    if drain:
                        # from an IDLE pipeline to a fully drained one: nothing is uploaded or queued before the clock
                        # starts, nothing is left in flight when it stops; every batch counted was delivered inside it
                        start.wait()
                        e_.prefetch(f_[3])
                        for it in range(steps_each):
                            if it + 1 < steps_each:
                                e_.prefetch(f_[3])
                            e_.process_device(f_[3], *f_[2])
                        return
                    if prefetch:
                        e_.prefetch(f_[3])
                        e_.prefetch(f_[3])
                        e_.process_device(f_[3], *f_[2])  # (one pipelined step before the clock starts)
                    start.wait()
                    for it in range(steps_each):
                        if prefetch:
                            e_.prefetch(f_[3])  # the next batch uploads under this one's kernels (ma_prefetch_batch)
                        e_.process_device(f_[3], *f_[2])

                ths = [threading.Thread(target=work, args=(e_, f_)) for e_, f_ in zip(engs, feeders)]
                for t_ in ths:
                    t_.start()
                start.wait()
                t_h = time.perf_counter()
                for t_ in ths:
                    t_.join()
                dt_ = time.perf_counter() - t_h
                for e_ in engs:
                    e_.close()
                os.environ.pop("MA_HBM_SHARE", None)
                in_mb = sum(x.nbytes for x in feeders[0][0].values()) / 1e6
                out_mb = sum(x.nbytes for d_ in feeders[0][1] for x in d_.values()) / 1e6
                return {"value": round(n_feeders * steps_each * hn / dt_, 2), "unit": "windows/s", "feeders": n_feeders,
                        "windows_per_batch": hn, "input_MB_per_batch": round(in_mb, 1), "output_MB_per_batch": round(out_mb, 1)}

            also["host_path"] = host_leg(1)
            also["host_path"]["note"] = ("MA_MEM_HOST with pinned caller buffers, ONE feeder thread and context: every lane uploads "
                                         "its slice and brings back packed records of what it wrote; the next batch is uploaded "
                                         "under this batch's kernels (ma_prefetch_batch); submitted windows/s, PCIe both ways "
                                         "inside the timed region")
            also["host_path"]["timing"] = "steady state over 5 batches: one batch uploaded and computing when the clock starts, one still in flight when it stops"
            also["host_path_start_to_drain"] = host_leg(1, steps_each=20, drain=True)
            also["host_path_start_to_drain"]["note"] = ("the same route timed from an idle pipeline to a drained one over 20 batches: "
                                                        "every counted batch was uploaded, computed and delivered inside the clock")
            also["host_path_no_prefetch"] = host_leg(1, prefetch=False)
            also["host_path_no_prefetch"]["note"] = "the same without ma_prefetch_batch: upload, kernels and download of a batch in turn"
            also["host_path_2_feeders"] = host_leg(2)
            also["host_path_2_feeders"]["note"] = ("two contexts on one device, one host thread each "
                                                   "(examples/host_driver.cpp --feeders 2), each prefetching its next batch")
        except Exception as exc:  # the host leg must never cost the headline
            also["host_path"] = {"error": str(exc)[:200]}

    if rank == 0:
        out = {
            "metric": "microassembly windows/sec (whole node)", "value": round(asm_wps, 2), "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/i32 (f64 statistics)", "data": "synthetic",
            "config": {"workload": WORKLOADS.get(args.config, args.config),
                       "windows_per_step_per_gpu": n, "distinct_windows": n0,
                       "value_counts": "assembled windows (SURVEY 8d); all windows submitted per second are in "
                                       "submitted_windows_per_s",
                       "submitted_windows_per_s": round(wps_all, 2),
                       "k_attempts_per_window": round(stats.get("window_attempts", 0) / steps / n, 3),
                       "str_windows": f"every {args.str_every}th window carries a 12-copy tandem repeat" if args.str_every else "none",
                       "harder_windows": (f"every {args.hard_every}th: a 30-80 base tandem duplication in the sample (cycle at k = 25), "
                                          f"every {args.hard_every}th: a 60-100 base low-complexity stretch, every "
                                          f"{2 * args.hard_every}th: a 200 base dispersed duplication of the reference (repeat gate)")
                       if args.hard_every else "none",
                       "soft_clipped_reads": args.softclip, "reads_with_N": args.nfrac,
                       "reads_without_hint": f"every {args.nohint_every}th read pair" if args.nohint_every else "none",
                       "reads_per_window": round(R, 1),
                       "assembled_fraction": round(assembled_all / (n * world), 4),
                       "repeat_gated_fraction": round(gated / n, 4),
                       "windows_with_capacity_overflow": overflowed, "haplotypes_per_assembled_window": round(H, 2),
                       "sharding": "static, one process per GPU, no collective",
                       "rank_windows": rank_windows,
                       "oversubscribed": (f"REHEARSAL: {world} ranks share {torch.cuda.device_count()} device(s), gloo barrier -- exercises "
                                          "the launch / sharding path, not a scaling measurement") if oversub else None,
                       "input_synthesis_s": round(t_gen, 1)},
            "roofline": roof, "roofline_valu": roof_valu, "roofline_top5": top5, "gcups": gcups, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_mt,
            "parity_sample": parity,
            "step_algorithmic": {"MB_per_step": round(step_bytes / 1e6, 1),
                                 "GB_per_s": round(step_bytes / (elapsed / args.steps) / 1e9, 1),
                                 "frac_of_hbm_peak": round(step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
                                 "copy_ceiling_GBps": copy_ceiling,
                                 "frac_of_copy_ceiling": round(step_bytes / (elapsed / args.steps) / 1e9 / copy_ceiling, 5) if copy_ceiling else None},
            "build": build_stamp(),  # what this run loaded: tools/pmc_per_kernel.py stamps the counters of a PMC pass with it
            "kernel_ms_per_step": kernel_ms_per_step, "stages": stages,
            "work": {"pairs_per_window": round(pairs_w, 1), "dp_pairs_per_window": round(dp_w, 1),
                     "dp_pairs_by_region_width": {k_: round(stats.get(k_, 0) / steps / n, 2)
                                                  for k_ in ("dp_w41", "dp_w65", "dp_w129", "dp_wide")},
                     "kmer_instances_per_attempt": round(st["N_inst"], 1), "distinct_kmers_per_attempt": round(st["N_raw"], 1),
                     "nodes_after_lowcov_per_attempt": round(wstats.get("nodes_after_lowcov", 0) / attempts, 1),
                     "slow_instances_per_attempt": round(wstats.get("slow_instances", 0) / attempts, 1),
                     "mean_haplotype_len": round(L, 1), "variant_bases_per_assembled_window": round(var_bases, 1),
                     "mean_read_len": round(read_len, 1)},
            # SURVEY 8 f3 (next row), outside the metric's timed region: ma_annotate_batch over the same batch
            "also": also or None,
            "annotation": {"ms_per_step": round(annot_ms, 3), "variants_per_window": round(float(nvars.sum()) / n, 2),
                           "kernel_ms_per_step": {k_: round(v_, 3) for k_, v_ in annot_k.items()}},
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if rank == 0 and parity and parity["mismatches"]:
        sys.exit(3)  # a fast step whose results differ from the oracle's is not a result


if __name__ == "__main__":
    main()
