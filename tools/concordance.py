#!/usr/bin/env python3
"""Call-level concordance of the engine's records with a Lancet2 VCF.

The engine's outputs are window-relative; a maintainer who has a built Lancet2 runs both on the same windows and compares
CALLS -- the level at which the two third-party aligners the reference links (SPOA, minimap2) stop mattering:

  python tools/concordance.py --records engine.tsv --windows windows.tsv --vcf lancet2.vcf[.gz] [--qual-tol 1e-3]

  engine.tsv   what examples/host_driver.cpp --out writes: window, pos (0-based, window-relative), REF, ALT[,ALT..], QUAL, AD...
  windows.tsv  window index -> CHROM, START (0-based genome coordinate of the window's first base); one line per window
  lancet2.vcf  the reference's calls for the same windows (CHROM POS ID REF ALT QUAL ...; POS is 1-based)

Prints the number of calls both have, the ones only one side has, and the QUAL differences of the shared ones.  Exit code 0
when the call sets are equal and every shared QUAL agrees within the tolerance.  No Lancet2 binary can be built in the
build environment of this repository, so the tool ships with a self-test instead of a result (`--self-test`).
"""
import argparse
import gzip
import sys


def read_records(path):
    calls = {}
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        f = line.rstrip("\n").split("\t")
        calls[(int(f[0]), int(f[1]), f[2], f[3])] = float(f[4])
    return calls


def read_windows(path):
    win = {}
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        f = line.split()
        win[int(f[0])] = (f[1], int(f[2]))
    return win


def read_vcf(path):
    calls = {}
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as fh:
        for line in fh:
            if line.startswith("#"):
                continue
            f = line.rstrip("\n").split("\t")
            calls[(f[0], int(f[1]), f[3], f[4])] = float(f[5]) if f[5] != "." else float("nan")
    return calls


def compare(engine, windows, vcf, qual_tol):
    mine = {}
    for (w, pos, ref, alt), qual in engine.items():
        chrom, start = windows[w]
        mine[(chrom, start + pos + 1, ref, alt)] = qual  # VCF POS is 1-based
    both = sorted(set(mine) & set(vcf))
    only_engine = sorted(set(mine) - set(vcf))
    only_vcf = sorted(set(vcf) - set(mine))
    qual_off = [(k, mine[k], vcf[k]) for k in both if not (abs(mine[k] - vcf[k]) <= qual_tol * max(1.0, abs(vcf[k])))]
    return both, only_engine, only_vcf, qual_off


def report(both, only_engine, only_vcf, qual_off, out=sys.stdout):
    print(f"shared calls {len(both)}, engine only {len(only_engine)}, VCF only {len(only_vcf)}, QUAL beyond tolerance {len(qual_off)}", file=out)
    for tag, rows in (("engine only", only_engine), ("VCF only", only_vcf)):
        for k in rows[:20]:
            print(f"  {tag}: {k[0]}:{k[1]} {k[2]}>{k[3]}", file=out)
    for k, a, b in qual_off[:20]:
        print(f"  QUAL {k[0]}:{k[1]} {k[2]}>{k[3]}: engine {a:.4f} VCF {b:.4f}", file=out)
    return 0 if not (only_engine or only_vcf or qual_off) else 1


def self_test():
    engine = {(0, 10, "A", "C"): 30.0, (0, 40, "AT", "A"): 12.5, (1, 5, "G", "GTT,GT"): 7.0}
    windows = {0: ("chr1", 1000), 1: ("chr1", 2001)}
    vcf = {("chr1", 1011, "A", "C"): 30.0, ("chr1", 1041, "AT", "A"): 12.5001, ("chr1", 2007, "G", "GTT,GT"): 7.0}
    assert report(*compare(engine, windows, vcf, 1e-3), out=open("/dev/null", "w")) == 0
    vcf[("chr1", 3000, "T", "G")] = 5.0
    del vcf[("chr1", 1011, "A", "C")]
    vcf[("chr1", 1041, "AT", "A")] = 20.0
    both, oe, ov, qo = compare(engine, windows, vcf, 1e-3)
    assert (len(both), len(oe), len(ov), len(qo)) == (2, 1, 1, 1)
    print("self-test ok")
    return 0


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--records")
    ap.add_argument("--windows")
    ap.add_argument("--vcf")
    ap.add_argument("--qual-tol", type=float, default=1e-3)
    ap.add_argument("--self-test", action="store_true")
    a = ap.parse_args()
    if a.self_test:
        return self_test()
    if not (a.records and a.windows and a.vcf):
        ap.error("--records, --windows and --vcf are required")
    return report(*compare(read_records(a.records), read_windows(a.windows), read_vcf(a.vcf), a.qual_tol))


if __name__ == "__main__":
    sys.exit(main())
