"""Thin Python binding over the C-ABI (libmicroasm.so) -- plumbing only.

The compute lives in the HIP library.  There is NO CPU fallback: constructing an Engine without the
built library or without a HIP device raises.
"""
import ctypes as C
import os

import numpy as np

from . import capi


class EngineError(RuntimeError):
    pass


def load_library(path=None):
    path = path or capi.LIB_PATH
    if not os.path.exists(path):
        raise EngineError(f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the engine has no CPU fallback)")
    lib = capi.load_cdll(path)
    lib.ma_last_error.restype = C.c_char_p
    lib.ma_create.argtypes = [C.POINTER(capi.Params), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.ma_destroy.argtypes = [C.c_void_p]
    lib.ma_last_error.argtypes = [C.c_void_p]
    lib.ma_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.ma_synchronize.argtypes = [C.c_void_p]
    lib.ma_timing_control.argtypes = [C.c_void_p, C.c_int]
    lib.ma_repeat_gate_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ma_assemble_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ma_msa_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ma_genotype_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ma_process_batch.argtypes = [C.c_void_p] * 6
    lib.ma_prefetch_batch.argtypes = [C.c_void_p, C.c_void_p]
    lib.ma_annotate_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
    lib.ma_last_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
    lib.ma_last_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    lib.ma_set_streams.argtypes = [C.c_void_p, C.c_int]
    return lib


class Engine:
    """Mirrors the per-thread VariantBuilder of the reference (core/variant_builder.h:26-125): one
    Engine per GPU/stream, reused across batches of windows."""

    def __init__(self, params=None, device=0, memspace=capi.MA_MEM_HOST, lib_path=None):
        self.lib = load_library(lib_path)
        self.p = params or capi.default_params()
        self.memspace = memspace
        h = C.c_void_p()
        rc = self.lib.ma_create(C.byref(self.p), device, memspace, C.byref(h))
        if rc != 0:
            raise EngineError(f"ma_create failed with {rc} (-2 = no HIP device; the engine has no CPU fallback)")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.ma_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise EngineError(f"{what} failed ({rc}): {self.lib.ma_last_error(self.h).decode()}")

    def set_stream(self, stream_ptr):
        self._check(self.lib.ma_set_stream(self.h, C.c_void_p(stream_ptr)), "ma_set_stream")

    def synchronize(self):
        self._check(self.lib.ma_synchronize(self.h), "ma_synchronize")

    def timing_control(self, mode):
        self._check(self.lib.ma_timing_control(self.h, mode), "ma_timing_control")

    def kernel_times(self, cap=8192):
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        n = self.lib.ma_last_kernel_times(self.h, names, ms, cap)
        return [(names[i].decode(), float(ms[i])) for i in range(max(n, 0))]

    def set_streams(self, n):
        """Number of concurrent window ranges ma_process_batch uses (0 = automatic)."""
        self._check(self.lib.ma_set_streams(self.h, n), "ma_set_streams")

    def stats(self):
        """Work counters of include/microasm.h:ma_last_stats."""
        buf = (C.c_ulonglong * 20)()
        n = self.lib.ma_last_stats(self.h, buf, 20)
        keys = ("pairs", "dp_pairs", "windows", "window_attempts", "dp_w41", "dp_w65", "dp_w129", "dp_wide",
                "distinct_kmers", "nodes_after_lowcov", "slow_instances", "kmer_instances", "edge_queue", "count_queue",
                "aln_dp_cells", "poa_band_cells", "poa_band_fills", "poa_full_cells", "poa_alignments", "poa_direct_alignments")
        return {k: int(buf[i]) for i, k in enumerate(keys) if i < n}

    # ---- host-array convenience (MA_MEM_HOST): numpy in, numpy out ----
    def _alloc(self, spec):
        if self.memspace != capi.MA_MEM_HOST:
            raise EngineError("numpy convenience calls need MA_MEM_HOST")
        return capi.alloc_host(spec)

    def gate(self, arrs, n, nr):
        out = self._alloc(capi.gate_out_spec(n))
        b = capi.make_batch_struct(arrs, n, nr)
        o = capi.fill_struct(capi.GateOut, out)
        self._check(self.lib.ma_repeat_gate_batch(self.h, C.byref(b), C.byref(o)), "ma_repeat_gate_batch")
        return out

    def assemble(self, arrs, n, nr):
        out = self._alloc(capi.asm_out_spec(self.p, n))
        b = capi.make_batch_struct(arrs, n, nr)
        o = capi.fill_struct(capi.AsmOut, out)
        self._check(self.lib.ma_assemble_batch(self.h, C.byref(b), C.byref(o)), "ma_assemble_batch")
        return out

    def msa(self, arrs, n, nr, asm):
        out = self._alloc(capi.var_out_spec(self.p, n))
        b = capi.make_batch_struct(arrs, n, nr)
        a = capi.fill_struct(capi.AsmOut, asm)
        o = capi.fill_struct(capi.VarOut, out)
        self._check(self.lib.ma_msa_batch(self.h, C.byref(b), C.byref(a), C.byref(o)), "ma_msa_batch")
        return out

    def genotype(self, arrs, n, nr, asm, var, debug=True):
        out = self._alloc(capi.geno_out_spec(self.p, n, nr, debug))
        b = capi.make_batch_struct(arrs, n, nr)
        a = capi.fill_struct(capi.AsmOut, asm)
        v = capi.fill_struct(capi.VarOut, var)
        o = capi.fill_struct(capi.GenoOut, out)
        self._check(self.lib.ma_genotype_batch(self.h, C.byref(b), C.byref(a), C.byref(v), C.byref(o)),
                    "ma_genotype_batch")
        return out

    def annotate(self, arrs, n, nr, asm, var, gc_frac=0.41):
        """VariantAnnotator (core/variant_annotator.cpp:43-101): SEQ_CX + GRAPH_CX of every variant."""
        out = self._alloc(capi.cx_out_spec(self.p, n))
        b = capi.make_batch_struct(arrs, n, nr)
        a = capi.fill_struct(capi.AsmOut, asm)
        v = capi.fill_struct(capi.VarOut, var)
        o = capi.fill_struct(capi.CxOut, out)
        self._check(self.lib.ma_annotate_batch(self.h, C.byref(b), C.byref(a), C.byref(v), C.c_double(gc_frac),
                                               C.byref(o)), "ma_annotate_batch")
        return out

    def process(self, arrs, n, nr, debug=False):
        g = self._alloc(capi.gate_out_spec(n))
        a = self._alloc(capi.asm_out_spec(self.p, n))
        v = self._alloc(capi.var_out_spec(self.p, n))
        q = self._alloc(capi.geno_out_spec(self.p, n, nr, debug))
        b = capi.make_batch_struct(arrs, n, nr)
        self._check(self.lib.ma_process_batch(self.h, C.byref(b), C.byref(capi.fill_struct(capi.GateOut, g)),
                                              C.byref(capi.fill_struct(capi.AsmOut, a)),
                                              C.byref(capi.fill_struct(capi.VarOut, v)),
                                              C.byref(capi.fill_struct(capi.GenoOut, q))), "ma_process_batch")
        return g, a, v, q

    # ---- device-resident path (MA_MEM_DEVICE): dicts of torch tensors (or raw int pointers) ----
    def process_device(self, batch_struct, gate, asm, var, geno):
        self._check(self.lib.ma_process_batch(self.h, C.byref(batch_struct), C.byref(gate), C.byref(asm),
                                              C.byref(var), C.byref(geno)), "ma_process_batch")

    def prefetch(self, batch_struct):
        """MA_MEM_HOST: start uploading the batch that the next process_device() call will be given (ma_prefetch_batch)."""
        self._check(self.lib.ma_prefetch_batch(self.h, C.byref(batch_struct)), "ma_prefetch_batch")

    def annotate_device(self, batch_struct, asm, var, cx, gc_frac=0.41):
        self._check(self.lib.ma_annotate_batch(self.h, C.byref(batch_struct), C.byref(asm), C.byref(var),
                                               C.c_double(gc_frac), C.byref(cx)), "ma_annotate_batch")
