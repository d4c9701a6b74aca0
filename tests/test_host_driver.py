"""examples/host_driver.cpp -- the C++ host a Lancet2 maintainer would write on top of the C-ABI:
Flatten(), one context + feeder thread per device, static sharding of batches, ordered flush.  CPU: it compiles against
include/microasm.h with plain g++ and fails loudly without a device (no CPU fallback).  GPU: its records equal what the
oracle says about the very batch it flattened."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from lancet2_amd import capi

REPO = capi.REPO
SRC = os.path.join(REPO, "examples", "host_driver.cpp")


def build_driver(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "host_driver")
    libdir = os.path.join(REPO, "lancet2_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", SRC, "-I", os.path.join(REPO, "include"), "-L", libdir, "-lmicroasm",
                           f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined", "-lpthread", "-o", exe])
    return exe


def test_host_driver_builds_and_has_no_cpu_fallback(tmp_path):
    exe = build_driver(tmp_path)
    have_gpu = os.path.exists("/dev/kfd")
    r = subprocess.run([exe, "--windows", "2", "--batch", "2"], capture_output=True, text=True)
    if not have_gpu:
        assert r.returncode == 3 and "no CPU fallback" in r.stderr, (r.returncode, r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["somatic", "germline"])
def test_host_driver_records_match_the_oracle(tmp_path, mode):
    from harness import OracleEngine, variants_of
    exe = build_driver(tmp_path)
    dump = tmp_path / "batch"
    dump.mkdir()
    out = tmp_path / "records.tsv"
    n = 24
    # (somatic: four feeder threads = four contexts on device 0, the multi-context path of an 8-GPU host in miniature)
    cmd = [exe, "--windows", str(n), "--batch", "5", "--devices", "1", "--feeders", "4" if mode == "somatic" else "3",
           "--dump", str(dump), "--out", str(out)]
    if mode == "germline":
        cmd.append("--germline")
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    dt = {"u8": np.uint8, "u32": np.uint32, "u64": np.uint64, "i32": np.int32}
    arrs = {f.rsplit(".", 1)[0]: np.fromfile(str(dump / f), dtype=dt[f.rsplit(".", 1)[1]]) for f in os.listdir(dump)}
    nr = len(arrs["read_qname_id"])
    params = capi.default_params(min_k=25, max_k=25, case_ctrl_mode=0 if mode == "germline" else 1)
    orc = OracleEngine(params)
    a = orc.assemble(arrs, n, nr)
    v = orc.msa(arrs, n, nr, a)
    q = orc.genotype(arrs, n, nr, a, v, debug=False)
    want = []
    NA = params.max_alts + 1
    for w in range(n):
        for vx, (pos, ref, alts) in enumerate(variants_of(params, v, w)):
            vi = w * params.max_vars + vx
            ads = []
            for s in range(params.num_samples):
                c = q["allele_counts"][(vi * params.num_samples + s) * NA * 2:(vi * params.num_samples + s + 1) * NA * 2].reshape(NA, 2).sum(axis=1)
                ads.append(",".join(str(int(x)) for x in c[:len(alts) + 1]))
            want.append((w, pos, ref.decode(), ",".join(x.decode() for x in alts), float(q["var_qual"][vi]), ads))
    got = []
    for line in open(out):
        if line.startswith("#"):
            continue
        f = line.rstrip("\n").split("\t")
        got.append((int(f[0]), int(f[1]), f[2], f[3], float(f[4]), f[5:]))
    assert len(got) == len(want) and len(got) >= n  # ordered flush: window order, one SNV + one deletion planted per window
    for g_, w_ in zip(got, want):
        assert g_[:4] == w_[:4] and g_[5] == w_[5], (g_, w_)
        assert abs(g_[4] - w_[4]) <= 1e-5
