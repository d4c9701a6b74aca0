"""Provenance stamp of the kernel sources: profiles/*.json made from PMC passes carry it, bench.py compares it with the
sources it runs, so that counters of an older build are not priced against a newer kernel's time without saying so."""
import glob
import hashlib
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(REPO, "lancet2_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(REPO, "lancet2_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(REPO, "include", "*.h")) + glob.glob(os.path.join(REPO, "include", "*.inc")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def lib_sha16():
    """hash of the built library bench.py / the profiler actually load (None if it is not built)"""
    path = os.path.join(REPO, "lancet2_amd", "libmicroasm.so")
    if not os.path.exists(path):
        return None
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for chunk in iter(lambda: fh.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()[:16]


def build_stamp():
    return {"csrc_sha16": csrc_sha16(), "lib_sha16": lib_sha16()}
