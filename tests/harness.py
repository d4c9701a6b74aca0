"""Test harness: loads the ORACLE (tests are the only place allowed to) and the product library,
and runs the batched stage entry points of either through identical numpy buffers."""
import ctypes as C
import os
import subprocess

import numpy as np

from lancet2_amd import capi

REPO = capi.REPO
ORACLE_DIR = os.path.join(REPO, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")


def load_oracle():
    alt = os.environ.get("MA_ORACLE_LIB")  # bench.py's cpu_baseline leg: the -O3 -march=native build of the same sources
    if alt and os.path.exists(alt):
        return _declare(C.CDLL(alt))
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".hpp"))]
    if (not os.path.exists(ORACLE_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_LIB) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return _declare(C.CDLL(ORACLE_LIB))


def _declare(lib):
    lib.orc_hamming.restype = C.c_uint64
    lib.orc_hash64.restype = C.c_uint64
    lib.orc_phred.restype = C.c_double
    lib.orc_median_u32.restype = C.c_uint32
    lib.orc_edit_distance.restype = C.c_uint32
    lib.orc_refpos_to_qpos.restype = C.c_uint64
    lib.orc_entropy.restype = C.c_float
    lib.orc_longdust.restype = C.c_double
    return lib


_ORACLE = None


def oracle():
    global _ORACLE
    if _ORACLE is None:
        _ORACLE = load_oracle()
    return _ORACLE


class OracleEngine:
    """Runs the oracle's batched stages on host numpy arrays."""

    def __init__(self, params):
        self.p = params
        self.lib = oracle()

    def gate(self, arrs, n, nr):
        out = capi.alloc_host(capi.gate_out_spec(n))
        b = capi.make_batch_struct(arrs, n, nr)
        rc = self.lib.orc_repeat_gate_batch(C.byref(self.p), C.byref(b), C.byref(capi.fill_struct(capi.GateOut, out)))
        assert rc == 0
        return out

    def assemble(self, arrs, n, nr):
        out = capi.alloc_host(capi.asm_out_spec(self.p, n))
        b = capi.make_batch_struct(arrs, n, nr)
        rc = self.lib.orc_assemble_batch(C.byref(self.p), C.byref(b), C.byref(capi.fill_struct(capi.AsmOut, out)))
        assert rc == 0
        return out

    def msa(self, arrs, n, nr, asm):
        out = capi.alloc_host(capi.var_out_spec(self.p, n))
        b = capi.make_batch_struct(arrs, n, nr)
        rc = self.lib.orc_msa_batch(C.byref(self.p), C.byref(b), C.byref(capi.fill_struct(capi.AsmOut, asm)),
                                    C.byref(capi.fill_struct(capi.VarOut, out)))
        assert rc == 0
        return out

    def genotype(self, arrs, n, nr, asm, var, debug=True):
        out = capi.alloc_host(capi.geno_out_spec(self.p, n, nr, debug))
        b = capi.make_batch_struct(arrs, n, nr)
        rc = self.lib.orc_genotype_batch(C.byref(self.p), C.byref(b), C.byref(capi.fill_struct(capi.AsmOut, asm)),
                                         C.byref(capi.fill_struct(capi.VarOut, var)),
                                         C.byref(capi.fill_struct(capi.GenoOut, out)))
        assert rc == 0
        return out


    def annotate(self, arrs, n, nr, asm, var, gc_frac=0.41):
        out = capi.alloc_host(capi.cx_out_spec(self.p, n))
        b = capi.make_batch_struct(arrs, n, nr)
        rc = self.lib.orc_annotate_batch(C.byref(self.p), C.byref(b), C.byref(capi.fill_struct(capi.AsmOut, asm)),
                                         C.byref(capi.fill_struct(capi.VarOut, var)), C.c_double(gc_frac),
                                         C.byref(capi.fill_struct(capi.CxOut, out)))
        assert rc == 0
        return out


def haplotypes_of(p, asm, w):
    """-> list over components of list of haplotype byte strings."""
    res = []
    for c in range(int(asm["win_ncomp"][w])):
        ci = w * p.max_comps + c
        hs = []
        for h in range(int(asm["comp_nhaps"][ci])):
            hi = w * p.max_haps + int(asm["comp_hap0"][ci]) + h
            hs.append(bytes(asm["hap_bases"][hi * p.max_hap_len: hi * p.max_hap_len + int(asm["hap_len"][hi])]))
        res.append(hs)
    return res


def variants_of(p, var, w):
    res = []
    pool = var["allele_pool"][w * p.max_allele_bytes:(w + 1) * p.max_allele_bytes]
    for v in range(int(var["win_nvars"][w])):
        vi = w * p.max_vars + v
        ref = bytes(pool[int(var["var_ref_off"][vi]): int(var["var_ref_off"][vi]) + int(var["var_ref_len"][vi])])
        alts = []
        for a in range(int(var["var_nalts"][vi])):
            ai = vi * p.max_alts + a
            alts.append(bytes(pool[int(var["alt_off"][ai]): int(var["alt_off"][ai]) + int(var["alt_len"][ai])]))
        res.append((int(var["var_pos"][vi]), ref, tuple(alts)))
    return res


def compare_asm(p, got, want, n):
    """Bit-exact comparison of the used part of two ma_asm_out_t buffers; returns list of mismatch strings."""
    bad = []
    for name in ("win_status", "win_k", "win_ncomp"):
        if not np.array_equal(got[name], want[name]):
            idx = np.nonzero(got[name] != want[name])[0][:5]
            bad.append(f"{name} differs at windows {idx.tolist()}: got {got[name][idx].tolist()} want {want[name][idx].tolist()}")
    if bad:
        return bad
    MC, MH, ML, MR = p.max_comps, p.max_haps, p.max_hap_len, p.max_runs
    for w in range(n):
        for c in range(int(want["win_ncomp"][w])):
            ci = w * MC + c
            for name, width in (("comp_anchor", 1), ("comp_hap0", 1), ("comp_nhaps", 1), ("comp_cx", 3), ("comp_cxf", 4)):
                g, x = got[name][ci * width:(ci + 1) * width], want[name][ci * width:(ci + 1) * width]
                if not np.array_equal(g.view(np.uint8), x.view(np.uint8)):
                    bad.append(f"w{w} c{c} {name}: got {g.tolist()} want {x.tolist()}")
            for h in range(int(want["comp_nhaps"][ci])):
                hi = w * MH + int(want["comp_hap0"][ci]) + h
                if got["hap_len"][hi] != want["hap_len"][hi] or got["hap_nruns"][hi] != want["hap_nruns"][hi]:
                    bad.append(f"w{w} c{c} h{h} len/nruns: got {got['hap_len'][hi]},{got['hap_nruns'][hi]} want {want['hap_len'][hi]},{want['hap_nruns'][hi]}")
                    continue
                L, R = int(want["hap_len"][hi]), int(want["hap_nruns"][hi])
                if not np.array_equal(got["hap_bases"][hi * ML: hi * ML + L], want["hap_bases"][hi * ML: hi * ML + L]):
                    bad.append(f"w{w} c{c} h{h} bases differ")
                if not np.array_equal(got["hap_runs"][hi * MR * 2: (hi * MR + R) * 2], want["hap_runs"][hi * MR * 2: (hi * MR + R) * 2]):
                    bad.append(f"w{w} c{c} h{h} runs differ")
                gs, xs = got["hap_stats"][hi * 6:(hi + 1) * 6], want["hap_stats"][hi * 6:(hi + 1) * 6]
                if not np.array_equal(gs.view(np.uint64), xs.view(np.uint64)):
                    bad.append(f"w{w} c{c} h{h} stats: got {gs.tolist()} want {xs.tolist()}")
    return bad


def compare_vars(p, got, want, n):
    bad = []
    if not np.array_equal(got["win_nvars"], want["win_nvars"]):
        return [f"win_nvars got {got['win_nvars'].tolist()} want {want['win_nvars'].tolist()}"]
    MH, MV, MA, MP = p.max_haps, p.max_vars, p.max_alts, p.max_allele_bytes
    for w in range(n):
        nv = int(want["win_nvars"][w])
        for name, width in (("var_comp", 1), ("var_pos", 1), ("var_ref_start", 1), ("var_ref_off", 1), ("var_ref_len", 1),
                            ("var_nalts", 1), ("var_hap_allele", MH), ("var_hap_start", MH)):
            g = got[name][w * MV * width:(w * MV + nv) * width]
            x = want[name][w * MV * width:(w * MV + nv) * width]
            if not np.array_equal(g, x):
                bad.append(f"w{w} {name}: got {g.tolist()} want {x.tolist()}")
        for v in range(nv):
            vi = w * MV + v
            na = int(want["var_nalts"][vi])
            for name in ("alt_off", "alt_len", "alt_type", "alt_length"):
                g, x = got[name][vi * MA: vi * MA + na], want[name][vi * MA: vi * MA + na]
                if not np.array_equal(g, x):
                    bad.append(f"w{w} v{v} {name}: got {g.tolist()} want {x.tolist()}")
        if variants_of(p, got, w) != variants_of(p, want, w):
            bad.append(f"w{w} alleles: got {variants_of(p, got, w)} want {variants_of(p, want, w)}")
    return bad


def compare_geno(p, got, want, n, nr, win_nvars, read_win_off):
    bad = []
    MH, MV, MCG = p.max_haps, p.max_vars, p.max_cigar
    if not np.array_equal(got["aln_rec"], want["aln_rec"]):
        idx = np.nonzero(got["aln_rec"].reshape(-1, 6) != want["aln_rec"].reshape(-1, 6))[0]
        for i in np.unique(idx)[:8]:
            bad.append(f"aln_rec read {i // MH} slot {i % MH}: got {got['aln_rec'].reshape(-1, 6)[i].tolist()} "
                       f"want {want['aln_rec'].reshape(-1, 6)[i].tolist()}")
    gc, wc = got["aln_cigar"].reshape(-1, 1 + MCG), want["aln_cigar"].reshape(-1, 1 + MCG)
    if not np.array_equal(gc[:, 0], wc[:, 0]):
        bad.append("cigar op counts differ")
    else:
        for i in np.nonzero(wc[:, 0])[0]:
            k = min(int(wc[i, 0]), MCG)
            if not np.array_equal(gc[i, 1:1 + k], wc[i, 1:1 + k]):
                bad.append(f"cigar read {i // MH} slot {i % MH}: got {gc[i, 1:1 + k].tolist()} want {wc[i, 1:1 + k].tolist()}")
                if len(bad) > 10:
                    break
    if not np.array_equal(got["asg_allele"], want["asg_allele"]):
        i = np.nonzero(got["asg_allele"] != want["asg_allele"])[0][:8]
        bad.append(f"asg_allele differs at {i.tolist()}: got {got['asg_allele'][i].tolist()} want {want['asg_allele'][i].tolist()}")
    if not np.array_equal(got["asg_score"].view(np.uint64), want["asg_score"].view(np.uint64)):
        i = np.nonzero(got["asg_score"] != want["asg_score"])[0][:8]
        bad.append(f"asg_score differs at {i.tolist()}: got {got['asg_score'][i].tolist()} want {want['asg_score'][i].tolist()}")
    if not np.array_equal(got["allele_counts"], want["allele_counts"]):
        i = np.nonzero(got["allele_counts"] != want["allele_counts"])[0][:8]
        bad.append(f"allele_counts differ at {i.tolist()}: got {got['allele_counts'][i].tolist()} want {want['allele_counts'][i].tolist()}")
    for key in ("var_pl", "var_gq"):
        if key in want and key in got and not np.array_equal(got[key], want[key]):
            i = np.nonzero(got[key] != want[key])[0][:8]
            bad.append(f"{key} differs at {i.tolist()}: got {got[key][i].tolist()} want {want[key][i].tolist()}")
    dq = np.abs(got["var_qual"] - want["var_qual"]).max() if len(want["var_qual"]) else 0.0
    if dq > 1e-9:
        bad.append(f"var_qual max abs diff {dq}")
    return bad


def compare_geno_calls(got, want):
    """The call-level outputs of the genotype stage (what a caller that does not ask for the debug taps gets): allele
    depths, PL, GQ equal; QUAL within 1e-9."""
    bad = []
    for key in ("allele_counts", "var_pl", "var_gq"):
        if not np.array_equal(got[key], want[key]):
            i = np.nonzero(got[key] != want[key])[0][:8]
            bad.append(f"{key} differs at {i.tolist()}: got {got[key][i].tolist()} want {want[key][i].tolist()}")
    dq = np.abs(got["var_qual"] - want["var_qual"]).max() if len(want["var_qual"]) else 0.0
    if not dq <= 1e-9:
        bad.append(f"var_qual max abs diff {dq}")
    return bad


SEQCX_FLOAT_TOL = 1e-5  # north_star's floating-point tolerance; the integer features must be equal


def compare_cx(p, got, want, win_nvars):
    """Integer features bit-exact, float features within SEQCX_FLOAT_TOL, only over the variants that exist."""
    MV = p.max_vars
    n = len(win_nvars)
    mask = (np.arange(MV)[None, :] < np.asarray(win_nvars)[:, None]).reshape(-1)
    for key, width in (("seq_cx_i", 4), ("seq_cx_f", 4), ("seq_cx_d", 3), ("graph_cx", 3)):
        g = got[key].reshape(n * MV, width)[mask]
        w = want[key].reshape(n * MV, width)[mask]
        if key == "seq_cx_i":
            bad = np.argwhere(g != w)
            assert bad.size == 0, f"{key}: first mismatch at {bad[0]}: got {g[tuple(bad[0])]} want {w[tuple(bad[0])]}"
        else:
            assert np.all(np.isfinite(g)), key
            err = np.abs(g.astype(np.float64) - w.astype(np.float64))
            assert err.size == 0 or err.max() <= SEQCX_FLOAT_TOL, f"{key}: max abs error {err.max()}"
    return int(mask.sum())


def handmade_annotation_case(p, cases):
    """Builds assembly + variant output buffers by hand: one window per case, one component.
    case = dict(haps=[ref, alt1, ...], ref_pos, ref_len, alts=[(alt_len, {hap_idx: start})...], cx=(cc, bp, deg),
                cxf=(unitig_ratio, cov_cv, tip_ratio))"""
    n = len(cases)
    asm = capi.alloc_host(capi.asm_out_spec(p, n))
    var = capi.alloc_host(capi.var_out_spec(p, n))
    MC, MH, ML, MV, MA = p.max_comps, p.max_haps, p.max_hap_len, p.max_vars, p.max_alts
    for w, cs in enumerate(cases):
        asm["win_ncomp"][w] = 1
        asm["win_k"][w] = 25
        ci = w * MC
        asm["comp_hap0"][ci] = 0
        asm["comp_nhaps"][ci] = len(cs["haps"])
        asm["comp_cx"][ci * 3:ci * 3 + 3] = cs.get("cx", (3, 2, 2))
        asm["comp_cxf"][ci * 4:ci * 4 + 4] = tuple(cs.get("cxf", (0.5, 0.25, 0.125))) + (0.0,)
        for h, seq in enumerate(cs["haps"]):
            hi = w * MH + h
            asm["hap_len"][hi] = len(seq)
            asm["hap_bases"][hi * ML:hi * ML + len(seq)] = np.frombuffer(seq.encode(), np.uint8)
        var["win_nvars"][w] = 1
        vi = w * MV
        var["var_comp"][vi] = 0
        var["var_ref_start"][vi] = cs["ref_pos"]
        var["var_ref_len"][vi] = cs["ref_len"]
        var["var_nalts"][vi] = len(cs["alts"])
        var["var_hap_start"][vi * MH] = cs["ref_pos"]
        for ai, (alen, sites) in enumerate(cs["alts"]):
            var["alt_len"][vi * MA + ai] = alen
            for h, st in sites.items():
                var["var_hap_allele"][vi * MH + h] = ai + 1
                var["var_hap_start"][vi * MH + h] = st
    return asm, var



class DeviceArena:
    """hipMalloc / hipMemcpy through ctypes on the HIP runtime the product library is using (no torch import: the
    first import of torch on a cold box takes minutes).  Test infrastructure for the MA_MEM_DEVICE entry points."""

    def __init__(self):
        import ctypes as C
        import importlib.util
        import os
        path = "libamdhip64.so"
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.submodule_search_locations:
            cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
            if os.path.exists(cand):
                path = cand  # the runtime capi.load_cdll() made global
        self.C = C
        self.hip = C.CDLL(path, mode=C.RTLD_GLOBAL)
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.hip.hipFree.argtypes = [C.c_void_p]
        self.ptrs = []

    def _check(self, rc, what):
        assert rc == 0, f"{what} failed with hipError {rc}"

    def alloc(self, nbytes):
        p = self.C.c_void_p()
        self._check(self.hip.hipMalloc(self.C.byref(p), max(int(nbytes), 16)), "hipMalloc")
        self._check(self.hip.hipMemset(p, 0, max(int(nbytes), 16)), "hipMemset")
        self.ptrs.append(p)
        return int(p.value)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.alloc(arr.nbytes)
        if arr.nbytes:
            self._check(self.hip.hipMemcpy(p, arr.ctypes.data, arr.nbytes, 1), "hipMemcpy H2D")
        return p

    def upload_unaligned(self, arr, shift=3, guard=256, junk=0x5A):
        """arr at an odd offset inside a junk-filled allocation, nothing but junk before and behind it: what a caller's
        exactly sized sub-allocation looks like to a kernel that reads aligned words around its input"""
        arr = np.ascontiguousarray(arr)
        total = arr.nbytes + 2 * guard + shift
        p = self.C.c_void_p()
        self._check(self.hip.hipMalloc(self.C.byref(p), total), "hipMalloc")
        self._check(self.hip.hipMemset(p, junk, total), "hipMemset")
        self.ptrs.append(p)
        at = int(p.value) + guard + shift
        if arr.nbytes:
            self._check(self.hip.hipMemcpy(at, arr.ctypes.data, arr.nbytes, 1), "hipMemcpy H2D")
        return at

    def download(self, ptr, dtype, count):
        out = np.zeros(int(count), dtype=dtype)
        self._check(self.hip.hipDeviceSynchronize(), "hipDeviceSynchronize")
        if out.nbytes:
            self._check(self.hip.hipMemcpy(out.ctypes.data, ptr, out.nbytes, 2), "hipMemcpy D2H")
        return out

    def close(self):
        for p in self.ptrs:
            self.hip.hipFree(p)
        self.ptrs = []
