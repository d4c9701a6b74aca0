"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/microasm.h declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from lancet2_amd import capi

HEADER = os.path.join(capi.REPO, "include", "microasm.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ma_[a-z_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(capi.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = C.CDLL(capi.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"libmicroasm.so does not export {s}"


def test_struct_layout_matches_header():
    # ma_params_t is 20 int32; pointer structs are arrays of pointers
    assert C.sizeof(capi.Params) == 20 * 4
    assert C.sizeof(capi.AsmOut) == 13 * C.sizeof(C.c_void_p)
    assert C.sizeof(capi.VarOut) == 14 * C.sizeof(C.c_void_p)
    assert C.sizeof(capi.GenoOut) == 6 * C.sizeof(C.c_void_p)
    lib = C.CDLL(capi.LIB_PATH)
    p = capi.Params()
    lib.ma_default_params(C.byref(p))
    d = capi.default_params()
    for name, _ in capi.Params._fields_:
        assert getattr(p, name) == getattr(d, name), name


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = C.CDLL(capi.LIB_PATH)
    h = C.c_void_p()
    p = capi.default_params()
    rc = lib.ma_create(C.byref(p), 0, 0, C.byref(h))
    assert rc == -2 and not h.value  # MA_ERR_NO_DEVICE: the product never computes on the CPU
