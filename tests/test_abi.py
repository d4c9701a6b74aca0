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
    lib = capi.load_cdll()
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"libmicroasm.so does not export {s}"


def test_struct_layout_matches_header():
    # ma_params_t is 20 int32; pointer structs are arrays of pointers
    assert C.sizeof(capi.Params) == 20 * 4
    assert C.sizeof(capi.AsmOut) == 13 * C.sizeof(C.c_void_p)
    assert C.sizeof(capi.VarOut) == 14 * C.sizeof(C.c_void_p)
    assert C.sizeof(capi.GenoOut) == 8 * C.sizeof(C.c_void_p)
    assert C.sizeof(capi.CxOut) == 4 * C.sizeof(C.c_void_p)
    lib = capi.load_cdll()
    p = capi.Params()
    lib.ma_default_params(C.byref(p))
    d = capi.default_params()
    for name, _ in capi.Params._fields_:
        assert getattr(p, name) == getattr(d, name), name


def test_no_cpu_fallback_without_device():
    if os.path.exists("/dev/kfd"):  # (not torch.cuda.is_available(): no second HIP runtime in the test process)
        pytest.skip("GPU present")
    lib = capi.load_cdll()
    h = C.c_void_p()
    p = capi.default_params()
    rc = lib.ma_create(C.byref(p), 0, 0, C.byref(h))
    assert rc == -2 and not h.value  # MA_ERR_NO_DEVICE: the product never computes on the CPU


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: no code under lancet2_amd/ (sources, bindings, build recipe) may include,
    import, link or call it (comments may mention it)."""
    pkg = os.path.join(capi.REPO, "lancet2_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if not (f.endswith((".py", ".hip", ".h", ".inc")) or f == "Makefile"):
                continue
            txt = open(os.path.join(root, f), errors="ignore").read()
            if f.endswith(".py"):
                txt = re.sub(r'"""(?:.|\n)*?"""', "", txt)
                txt = re.sub(r"#.*", "", txt)
            else:
                txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
                txt = re.sub(r"//.*", "", txt)
                if f == "Makefile":
                    txt = re.sub(r"#.*", "", txt)
            assert not re.search(r"oracle|orc_[a-z_]+\s*\(|harness", txt), os.path.join(root, f)


@pytest.mark.gpu
def test_error_codes_on_bad_arguments():
    """Argument errors are reported as negative codes, never thrown (include/microasm.h conventions)."""
    import numpy as np
    from lancet2_amd import synth
    from lancet2_amd.engine import load_library
    lib = load_library()
    h = C.c_void_p()
    p = capi.default_params()
    assert lib.ma_create(None, 0, 0, C.byref(h)) == -1            # MA_ERR_ARG
    assert lib.ma_create(C.byref(p), 99, 0, C.byref(h)) == -2     # no such device
    bad = capi.default_params(min_k=12, max_k=25)                  # even k
    rc = lib.ma_create(C.byref(bad), 0, 0, C.byref(h))
    assert rc in (-5, -1) and not h.value                         # MA_ERR_PARAM
    rc = lib.ma_create(C.byref(p), 0, 0, C.byref(h))
    assert rc == 0 and h.value, (rc, h.value)
    try:
        arrs, n, nr = synth.make_config_batch("C1", 1, first_index=5)
        b = capi.make_batch_struct(arrs, n, nr)
        assert lib.ma_repeat_gate_batch(h, C.byref(b), None) == -1
        assert lib.ma_assemble_batch(h, None, None) == -1
        assert lib.ma_process_batch(h, C.byref(b), None, None, None, None) == -1
        assert lib.ma_set_streams(h, -3) != 0
        assert lib.ma_last_error(h) is not None
        # an empty batch is fine
        b0 = capi.make_batch_struct({k: v[:0] if k not in ("ref_off", "read_win_off", "read_off") else v[:1]
                                     for k, v in arrs.items()}, 0, 0)
        g = capi.alloc_host(capi.gate_out_spec(1))
        assert lib.ma_repeat_gate_batch(h, C.byref(b0), C.byref(capi.fill_struct(capi.GateOut, g))) == 0
    finally:
        lib.ma_destroy(h)


def test_python_constants_mirror_the_header():
    """The status bits and read flags of lancet2_amd/capi.py are the enum values of include/microasm.h."""
    import re
    from lancet2_amd import capi
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "microasm.h")).read()
    found = {}
    for name, expr in re.findall(r"\b(MA_[A-Z_0-9]+)\s*=\s*([^,/\n}]+)", text):
        expr = expr.strip()
        m = re.fullmatch(r"(\d+)u?\s*<<\s*(\d+)", expr)
        if m:
            found[name] = int(m.group(1)) << int(m.group(2))
        elif re.fullmatch(r"-?\d+u?", expr):
            found[name] = int(expr.rstrip("u"))
    assert {"MA_W_NO_HAPLOTYPE", "MA_W_TABLE_OVERFLOW", "MA_MEM_DEVICE"} <= set(found)
    checked = 0
    for name, value in found.items():
        if hasattr(capi, name):
            assert getattr(capi, name) == value, name
            checked += 1
    assert checked >= 8


def test_ctypes_offsets_match_the_compiled_header(tmp_path):
    """Every field of every boundary struct sits where a C compiler puts it: a probe built with gcc from
    include/microasm.h prints sizeof / offsetof, the ctypes mirror must agree field by field."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    structs = {"ma_params_t": capi.Params, "ma_batch_t": capi.Batch, "ma_gate_out_t": capi.GateOut,
               "ma_asm_out_t": capi.AsmOut, "ma_var_out_t": capi.VarOut, "ma_geno_out_t": capi.GenoOut,
               "ma_cx_out_t": capi.CxOut}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "microasm.h"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.dirname(HEADER), str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"
