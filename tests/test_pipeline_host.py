"""The host shell of the `pipeline` path: lancet2_amd/host/pipeline_host.hpp + examples/pipeline_driver.cpp.
CPU: the C++ unit checks (tiling, padding, comparator, active-region rules, downsampling, store) and that the driver
builds and refuses to run without a device.  GPU: a small genome + two SAM files through the whole driver -- window tiling,
gates, read collection, batches with ma_prefetch_batch, VariantStore, ordered flush -- against the oracle run on the very
batches the driver flattened, with the reference's de-duplication rule restated here in a few lines."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from lancet2_amd import capi

REPO = capi.REPO
BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def build(tmp_path, src, out, extra=()):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / out)
    libdir = os.path.join(REPO, "lancet2_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(REPO, src), "-I", os.path.join(REPO, "include"),
                           *extra, "-lpthread", "-o", exe])
    return exe, libdir


def test_host_units(tmp_path):
    exe, _ = build(tmp_path, "tests/host/host_units.cpp", "host_units")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "host units ok" in r.stdout, r.stderr


def driver(tmp_path):
    libdir = os.path.join(REPO, "lancet2_amd")
    return build(tmp_path, "examples/pipeline_driver.cpp", "pipeline_driver",
                 ["-L", libdir, "-lmicroasm", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined", "-DLANCET2_AMD_WITH_ZLIB", "-lz"])[0]


# ---- fixture: a genome, two haplotypes with planted variants, paired reads as SAM records with honest CIGAR / MD -------
def make_haplotype(genome, variants):
    """variants: (pos0, ref_len, alt bytes) on genome coordinates -> (bases, ref position of every base or -1 for inserted)"""
    seq, pos = [], []
    last = 0
    for p, rl, alt in sorted(variants):
        seq.append(genome[last:p]); pos.append(np.arange(last, p))
        a = np.frombuffer(alt, dtype=np.uint8)
        if rl == len(a):           # substitution: the bases stand on the reference positions they replace
            seq.append(a); pos.append(np.arange(p, p + rl))
        else:                      # indel: anchor-free, inserted bases have no reference position
            seq.append(a); pos.append(np.full(len(a), -1))
        last = p + rl
    seq.append(genome[last:]); pos.append(np.arange(last, len(genome)))
    return np.concatenate(seq), np.concatenate(pos)


def sam_fields(genome, hseq, hpos, i0, n):
    """read = hseq[i0:i0+n]; returns (pos1, cigar, md) of its alignment to the genome"""
    rp = hpos[i0:i0 + n]
    bases = hseq[i0:i0 + n]
    ops, md, run = [], "", 0
    aligned = np.nonzero(rp >= 0)[0]
    lead, trail = int(aligned[0]), n - 1 - int(aligned[-1])
    if lead:
        ops.append((lead, "S"))
    prev = None
    for k in range(lead, n - trail):
        if rp[k] < 0:
            ops.append((1, "I"))
            continue
        if prev is not None and rp[k] > prev + 1:
            gap = rp[k] - prev - 1
            ops.append((int(gap), "D"))
            md += str(run) + "^" + bytes(genome[prev + 1:rp[k]]).decode()
            run = 0
        ops.append((1, "M"))
        if bases[k] == genome[rp[k]]:
            run += 1
        else:
            md += str(run) + chr(genome[rp[k]])
            run = 0
        prev = rp[k]
    md += str(run)
    if trail:
        ops.append((trail, "S"))
    merged = []
    for ln, op in ops:
        if merged and merged[-1][1] == op:
            merged[-1][0] += ln
        else:
            merged.append([ln, op])
    return int(rp[aligned[0]]) + 1, "".join(f"{ln}{op}" for ln, op in merged), md


def write_fixture(d, seed=7, glen=6000):
    rng = np.random.default_rng(seed)
    genome = BASES[rng.integers(0, 4, glen)]
    sub = lambda p: bytes([int(BASES[(np.searchsorted(BASES, genome[p]) + 1) % 4])])  # noqa: E731
    # (899 and 3299 lie where two windows overlap: both assemble them, the store keeps one call)
    germ_a = [(899, 1, sub(899)), (2300, 3, b""), (3299, 1, sub(3299))]
    germ_b = [(1500, 0, b"TTGA"), (3299, 1, sub(3299)), (4700, 1, sub(4700))]
    som = [(1000, 1, sub(1000)), (3100, 1, sub(3100)), (5200, 2, b"")]  # tumour only, on haplotype A
    with open(os.path.join(d, "ref.fa"), "w") as f:
        f.write(">chr1 test\n")
        s = bytes(genome).decode()
        for i in range(0, glen, 60):
            f.write(s[i:i + 60] + "\n")
    haps = {"normal": [make_haplotype(genome, germ_a), make_haplotype(genome, germ_b)],
            "tumor": [make_haplotype(genome, germ_a + som), make_haplotype(genome, germ_b)]}
    for name, depth in (("normal", 34), ("tumor", 44)):
        recs = []
        nfrag = int(depth * glen / 300)
        for fi in range(nfrag):
            hseq, hpos = haps[name][int(rng.integers(0, 2))]
            ins = int(max(170, rng.normal(400, 40)))
            fs = int(rng.integers(0, len(hseq) - ins))
            mapq = 60 if rng.random() > 0.04 else 7
            for mate, (i0, rev) in enumerate(((fs, False), (fs + ins - 150, True))):
                seq = hseq[i0:i0 + 150].copy()
                q = np.clip(np.round(37 - 15 * (np.arange(150) / 149.0) ** 2 + rng.normal(0, 1.5, 150)), 2, 41).astype(np.uint8)
                if rev:
                    q = q[::-1].copy()
                err = rng.random(150) < np.power(10.0, -q.astype(np.float64) / 10.0)
                seq[err] = BASES[(np.searchsorted(BASES, seq[err]) + rng.integers(1, 4, int(err.sum()))) % 4]
                hs2 = hseq.copy()
                hs2[i0:i0 + 150] = seq
                pos1, cigar, md = sam_fields(genome, hs2, hpos, i0, 150)
                flag = 0x1 | 0x2 | (0x10 if rev else 0x20) | (0x40 if mate == 0 else 0x80)
                recs.append((pos1, f"{name[0]}{fi}", flag, mapq, cigar, bytes(seq).decode(), bytes(q + 33).decode(), md))
        recs.sort(key=lambda r: r[0])
        with open(os.path.join(d, name + ".sam"), "w") as f:
            f.write("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:%d\n" % glen)
            for pos1, qn, flag, mapq, cigar, seq, qual, md in recs:
                f.write(f"{qn}\t{flag}\tchr1\t{pos1}\t{mapq}\t{cigar}\t=\t{pos1}\t0\t{seq}\t{qual}\tMD:Z:{md}\n")
    return genome


def test_pipeline_driver_builds_and_has_no_cpu_fallback(tmp_path):
    exe = driver(tmp_path)
    write_fixture(str(tmp_path))
    r = subprocess.run([exe, "--reference", str(tmp_path / "ref.fa"), "--normal", str(tmp_path / "normal.sam"),
                        "--tumor", str(tmp_path / "tumor.sam")], capture_output=True, text=True)
    if not os.path.exists("/dev/kfd"):
        assert r.returncode == 3 and "no CPU fallback" in r.stderr, (r.returncode, r.stderr)


@pytest.mark.gpu
def test_pipeline_driver_reproduces_the_oracle_with_overlap_duplicates_merged(tmp_path):
    from harness import OracleEngine, variants_of
    exe = driver(tmp_path)
    write_fixture(str(tmp_path))
    dump = tmp_path / "dump"
    dump.mkdir()
    out = tmp_path / "calls.tsv"
    r = subprocess.run([exe, "--reference", str(tmp_path / "ref.fa"), "--normal", str(tmp_path / "normal.sam"),
                        "--tumor", str(tmp_path / "tumor.sam"), "--region", "chr1:1-6000", "--min-kmer", "25", "--max-kmer", "25",
                        "--batch-windows", "3", "--dump", str(dump), "--out", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "7 windows" in r.stderr, r.stderr  # 1001-base windows every 800 bases while start + 1000 <= 6000
    params = capi.default_params(min_k=25, max_k=25)
    dt = {"u8": np.uint8, "u32": np.uint32, "u64": np.uint64, "i32": np.int32}
    store = {}
    n_windows = n_duplicates = n_unsupported = 0
    starts = []
    for b in sorted(os.listdir(dump)):
        arrs = {f.rsplit(".", 1)[0]: np.fromfile(str(dump / b / f), dtype=dt[f.rsplit(".", 1)[1]]) for f in os.listdir(dump / b)}
        wins = arrs.pop("windows").reshape(-1, 4)
        n, nr = len(wins), len(arrs["read_qname_id"])
        n_windows += n
        orc = OracleEngine(params)
        a = orc.assemble(arrs, n, nr)
        v = orc.msa(arrs, n, nr, a)
        q = orc.genotype(arrs, n, nr, a, v, debug=False)
        NA = params.max_alts + 1
        for w in range(n):
            starts.append(int(wins[w, 1]))
            for vx, (pos, ref, alts) in enumerate(variants_of(params, v, w)):
                vi = w * params.max_vars + vx
                ads = []
                for s in range(params.num_samples):
                    c = q["allele_counts"][(vi * params.num_samples + s) * NA * 2:(vi * params.num_samples + s + 1) * NA * 2]
                    ads.append([int(x) for x in c.reshape(NA, 2).sum(axis=1)[:len(alts) + 1]])
                rec = (int(wins[w, 1]) + pos, ref.decode(), ",".join(x.decode() for x in alts), float(q["var_qual"][vi]), ads)
                key = (rec[0], rec[1])  # one chromosome: CHROM + POS + REF (variant_call.cpp:37)
                cov = sum(sum(x) for x in ads)
                # core/variant_builder.cpp:184-199 (CollectSupportedCalls) runs BEFORE the store: a genotyped variant without
                # ALT support in any sample never becomes a VariantCall, so it cannot shadow a supported duplicate
                if not any(sum(x[1:]) > 0 for x in ads):
                    n_unsupported += 1
                    continue
                # core/variant_store.cpp:31-42: a duplicate replaces the stored call only if it has MORE total coverage
                n_duplicates += key in store
                if key not in store or sum(sum(x) for x in store[key][4]) < cov:
                    store[key] = rec
    assert n_windows == 7 and starts == [1 + 800 * i for i in range(7)]
    want = [store[k] for k in sorted(store) if any(sum(x[1:]) > 0 for x in store[k][4])]
    got = []
    for line in open(out):
        f = line.rstrip("\n").split("\t")
        got.append((f[0], int(f[1]), f[2], f[3], float(f[4]), [[int(x) for x in g.split(",")] for g in f[5:]]))
    assert len(got) == len(want) and len(got) >= 6, (len(got), len(want))
    assert n_duplicates >= 2  # overlapping windows (200 shared bases) did see the same site twice: the merge was exercised
    for g_, w_ in zip(got, want):
        assert g_[0] == "chr1" and g_[1:4] == w_[:3] and g_[5] == w_[4], (g_, w_)
        assert abs(g_[4] - w_[3]) <= 1e-5
    assert [g_[1] for g_ in got] == sorted(g_[1] for g_ in got)  # coordinate order
    planted = {900, 1001, 3101, 3300, 4701}  # the SNVs of the fixture (1-based)
    assert planted <= {g_[1] for g_ in got}, sorted(g_[1] for g_ in got)
    # the same run as VCF text: the reference's record layout, INFO and FORMAT key; AD / DP / GT / PL consistent with the TSV
    vcf = tmp_path / "calls.vcf"
    r = subprocess.run([exe, "--reference", str(tmp_path / "ref.fa"), "--normal", str(tmp_path / "normal.sam"),
                        "--tumor", str(tmp_path / "tumor.sam"), "--region", "chr1:1-6000", "--min-kmer", "25", "--max-kmer", "25",
                        "--batch-windows", "3", "--out-vcf", str(vcf)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = open(vcf).read().splitlines()
    assert lines[0] == "##fileformat=VCFv4.5" and any(x.startswith("##contig=<ID=chr1,length=6000>") for x in lines)
    body = [x.split("\t") for x in lines if not x.startswith("#")]
    assert [x for x in lines if x.startswith("#CHROM")][0].split("\t")[9:] == ["normal", "tumor"]
    assert len(body) == len(got)
    key = "GT:AD:ADF:ADR:DP:RMQ:NPBQ:SB:SCA:FLD:RPCD:BQCD:MQCD:ASMD:SDFC:PRAD:PANG:CMLOD:FSSE:AHDD:HSE:PDCV:PL:GQ".split(":")
    somatic = 0
    for f, g_ in zip(body, got):
        assert f[0] == "chr1" and int(f[1]) == g_[1] and f[3] == g_[2] and f[4] == g_[3] and f[8] == ":".join(key)
        assert abs(float(f[5]) - g_[4]) <= 0.005 + 1e-9  # QUAL with two decimals
        info = f[7].split(";")
        assert info[0] in ("SHARED", "CTRL", "CASE", "NONE") and any(x.startswith("TYPE=") for x in info)
        assert any(x.startswith("SEQ_CX=") and x.count(",") == 10 for x in info)
        assert any(x.startswith("GRAPH_CX=") and x.count(",") == 2 for x in info)
        somatic += info[0] == "CASE"
        for s_idx, col in enumerate(f[9:]):
            vals = dict(zip(key, col.split(":")))
            if vals["GT"] == "./.":
                assert sum(g_[5][s_idx]) == 0
                continue
            ad = [int(x) for x in vals["AD"].split(",")]
            assert ad == g_[5][s_idx] and int(vals["DP"]) == sum(ad)
            adf, adr = [int(x) for x in vals["ADF"].split(",")], [int(x) for x in vals["ADR"].split(",")]
            assert [a + b for a, b in zip(adf, adr)] == ad
            pl = [int(x) for x in vals["PL"].split(",")]
            k = len(ad)
            assert len(pl) == k * (k + 1) // 2 and min(pl) == 0
            a1, a2 = (int(x) for x in vals["GT"].split("/"))
            assert pl[a2 * (a2 + 1) // 2 + a1] == 0 and 0 <= int(vals["GQ"]) <= 99
    assert somatic >= 2  # the tumour-only SNVs at 1001 and 3101


def sam_to_bam(sam_path, bam_path, with_index=False, block_bytes=60000, corrupt_record=None):
    """a minimal BAM writer (SAM spec 4.2; BGZF blocks = gzip members with the BC extra field): fixture for LoadBam.
    with_index: also writes bam_path + ".bai" (SAM spec 5.2: binning index + 16 kb linear index over virtual offsets)"""
    import struct
    import zlib
    refs, recs = [], []
    text = ""
    for line in open(sam_path):
        if line[0] == "@":
            text += line
            if line.startswith("@SQ"):
                f = dict(x.split(":", 1) for x in line.rstrip("\n").split("\t")[1:])
                refs.append((f["SN"], int(f["LN"])))
            continue
        recs.append(line.rstrip("\n").split("\t"))
    names = [r[0] for r in refs]
    out = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs)))
    for nm, ln in refs:
        out += struct.pack("<i", len(nm) + 1) + nm.encode() + b"\0" + struct.pack("<i", ln)
    code = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    ops = {c: i for i, c in enumerate("MIDNSHP=X")}
    rec_at = []  # (uncompressed offset, reference id, start, end) of every record, for the index
    for f in recs:
        qn, flag, rn, pos, mapq, cig, rnext, pnext, tlen, seq, qual = f[:11]
        cigar, num = [], ""
        for ch in cig:
            if ch.isdigit():
                num += ch
            else:
                cigar.append((int(num) << 4) | ops[ch]); num = ""
        rid = names.index(rn)
        mrid = rid if rnext == "=" else (-1 if rnext == "*" else names.index(rnext))
        packed = bytearray((len(seq) + 1) // 2)
        for i, ch in enumerate(seq):
            packed[i // 2] |= code[ch] << (0 if i % 2 else 4)
        l_seq = len(seq) if corrupt_record != len(rec_at) else (1 << 20)  # (a length field that lies about its block)
        body = struct.pack("<iiBBHHHiiii", rid, int(pos) - 1, len(qn) + 1, int(mapq), 4680, len(cigar), int(flag), l_seq,
                           mrid, int(pnext) - 1, int(tlen))
        body += qn.encode() + b"\0" + b"".join(struct.pack("<I", c) for c in cigar) + bytes(packed)
        body += bytes(ord(c) - 33 for c in qual)
        for tag in f[11:]:
            if tag[3] == "Z":
                body += tag[:2].encode() + b"Z" + tag[5:].encode() + b"\0"
        span = sum(n >> 4 for n in cigar if (n & 15) in (0, 2, 3, 7, 8)) or 1
        rec_at.append((len(out), rid, int(pos) - 1, int(pos) - 1 + span))
        out += struct.pack("<i", len(body)) + body
    cstart = []
    with open(bam_path, "wb") as fh:
        def block(data):
            cstart.append(fh.tell())
            comp = zlib.compressobj(6, zlib.DEFLATED, -15)
            cdata = comp.compress(bytes(data)) + comp.flush()
            bsize = len(cdata) + 25
            fh.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + cdata +
                     struct.pack("<II", zlib.crc32(bytes(data)) & 0xFFFFFFFF, len(data)))
        for i in range(0, len(out), block_bytes):
            block(out[i:i + block_bytes])
        block(b"")
    if not with_index:
        return len(cstart)

    def voff(u):  # (a record may start in one block and end in the next: BGZF allows it)
        return (cstart[u // block_bytes] << 16) | (u % block_bytes)

    def reg2bin(beg, end):
        end -= 1
        for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
            if beg >> shift == end >> shift:
                return base + (beg >> shift)
        return 0

    bins = [dict() for _ in refs]
    lin = [dict() for _ in refs]
    for k, (u, rid, beg, end) in enumerate(rec_at):
        if rid < 0:
            continue
        v0, v1 = voff(u), voff(rec_at[k + 1][0] if k + 1 < len(rec_at) else len(out))
        ch = bins[rid].setdefault(reg2bin(beg, end), [])
        if ch and ch[-1][1] == v0:
            ch[-1][1] = v1
        else:
            ch.append([v0, v1])
        for w in range(beg >> 14, ((end - 1) >> 14) + 1):
            lin[rid][w] = min(lin[rid].get(w, v0), v0)
    bai = bytearray(b"BAI\1" + struct.pack("<i", len(refs)))
    for rid in range(len(refs)):
        bai += struct.pack("<i", len(bins[rid]))
        for b_, chunks in sorted(bins[rid].items()):
            bai += struct.pack("<Ii", b_, len(chunks))
            for v0, v1 in chunks:
                bai += struct.pack("<QQ", v0, v1)
        n_intv = (max(lin[rid]) + 1) if lin[rid] else 0
        bai += struct.pack("<i", n_intv)
        last = 0
        for w in range(n_intv):
            last = lin[rid].get(w, last)
            bai += struct.pack("<Q", last)
    with open(bam_path + ".bai", "wb") as fh:
        fh.write(bytes(bai))
    return len(cstart)


def test_bam_and_sam_sources_feed_identical_batches(tmp_path):
    """The same alignments as SAM text and as BAM (BGZF through zlib): the extract stage flattens byte-identical batches
    (window tiling, gates, collector and Flatten run without a device: --extract-only)."""
    exe = driver(tmp_path)
    write_fixture(str(tmp_path))
    for name in ("normal", "tumor"):
        sam_to_bam(str(tmp_path / (name + ".sam")), str(tmp_path / (name + ".bam")))
    dumps = {}
    for ext in ("sam", "bam"):
        d = tmp_path / ("dump_" + ext)
        d.mkdir()
        # the sample name is the file's stem in both runs, so the comparator's sample-name key agrees
        r = subprocess.run([exe, "--reference", str(tmp_path / "ref.fa"), "--normal", str(tmp_path / ("normal." + ext)),
                            "--tumor", str(tmp_path / ("tumor." + ext)), "--region", "chr1:1-6000", "--batch-windows", "4",
                            "--dump", str(d), "--extract-only"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        dumps[ext] = {(b, f): open(d / b / f, "rb").read() for b in sorted(os.listdir(d)) for f in sorted(os.listdir(d / b))}
    assert dumps["sam"].keys() == dumps["bam"].keys() and len(dumps["sam"]) >= 22
    for k in dumps["sam"]:
        assert dumps["sam"][k] == dumps["bam"][k], k


def test_indexed_bam_reads_only_the_blocks_of_the_region(tmp_path):
    """BAM + .bai (BGZF virtual offsets, binning + linear index; no htslib): for whole-genome and sub-region runs the extract
    stage flattens byte-identical batches from SAM text, from the BAM read whole, and from the BAM through its index -- and
    for a sub-region the indexed run inflates a fraction of the file's blocks (core/read_collector.cpp:106-204 iterates
    htslib regions; round 3 loaded the whole file)."""
    exe = driver(tmp_path)
    write_fixture(str(tmp_path))
    nblocks = {}
    for name in ("normal", "tumor"):
        nblocks[name] = sam_to_bam(str(tmp_path / (name + ".sam")), str(tmp_path / (name + ".bam")), with_index=True, block_bytes=6000)

    def run(ext, region, tag, env=None):
        d = tmp_path / ("dump_" + tag)
        d.mkdir()
        e = dict(os.environ)
        e.update(env or {})
        r = subprocess.run([exe, "--reference", str(tmp_path / "ref.fa"), "--normal", str(tmp_path / ("normal." + ext)),
                            "--tumor", str(tmp_path / ("tumor." + ext)), "--batch-windows", "4", "--dump", str(d), "--extract-only"] +
                           (["--region", region] if region else []), capture_output=True, text=True, env=e)
        assert r.returncode == 0, r.stderr
        files = {(b, f): open(d / b / f, "rb").read() for b in sorted(os.listdir(d)) for f in sorted(os.listdir(d / b))}
        return files, r.stderr

    for region, tag in ((None, "all"), ("chr1:2400-3900", "sub"), ("chr1:1-900", "head"), ("chr1:5200-6000", "tail")):
        sam, _ = run("sam", region, tag + "_sam")
        whole, _ = run("bam", region, tag + "_whole", {"PIPELINE_NO_INDEX": "1"})
        idx, log = run("bam", region, tag + "_idx")
        assert sam.keys() == whole.keys() == idx.keys() and len(sam) >= 11, (tag, len(sam), len(idx))
        for k in sam:
            assert sam[k] == whole[k] == idx[k], (tag, k)
        assert "indexed BAM" in log and "extract" in log and "windows/s" in log, log
        inflated = int(log.split("indexed BAM: ")[1].split(" BGZF")[0])
        if tag == "sub":  # ~1.5 kb of a 6 kb genome: well under the whole file (re-reads of a block by neighbouring windows count)
            assert 0 < inflated < (nblocks["normal"] + nblocks["tumor"]) * 3, (inflated, nblocks)


def test_bed_file_gives_the_regions(tmp_path):
    """--bed-file (core/bed_parser.cpp:27-96): three tab-separated columns, the span taken as it is; the same windows as the
    equivalent --region arguments; a malformed line is an error."""
    exe = driver(tmp_path)
    write_fixture(str(tmp_path))
    bed = tmp_path / "targets.bed"
    bed.write_text("# panel\nchr1\t1000\t2400\n\nchr1\t4000\t5200\n")
    outs = {}
    for tag, extra in (("bed", ["--bed-file", str(bed)]), ("regions", ["--region", "chr1:1000-2400", "--region", "chr1:4000-5200"])):
        d = tmp_path / ("dump_" + tag)
        d.mkdir()
        r = subprocess.run([exe, "--reference", str(tmp_path / "ref.fa"), "--normal", str(tmp_path / "normal.sam"),
                            "--tumor", str(tmp_path / "tumor.sam"), "--batch-windows", "4", "--dump", str(d), "--extract-only"] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs[tag] = {(b, f): open(d / b / f, "rb").read() for b in sorted(os.listdir(d)) for f in sorted(os.listdir(d / b))}
    assert outs["bed"] == outs["regions"] and len(outs["bed"]) >= 11
    bad = tmp_path / "bad.bed"
    bad.write_text("chr1\t100\n")
    r = subprocess.run([exe, "--reference", str(tmp_path / "ref.fa"), "--normal", str(tmp_path / "normal.sam"), "--bed-file", str(bad),
                        "--extract-only"], capture_output=True, text=True)
    assert r.returncode == 2 and "Invalid bed line with 2 columns at line number 1" in r.stderr, r.stderr
    bad.write_text("chrZ\t100\t200\n")
    r = subprocess.run([exe, "--reference", str(tmp_path / "ref.fa"), "--normal", str(tmp_path / "normal.sam"), "--bed-file", str(bad),
                        "--extract-only"], capture_output=True, text=True)
    assert r.returncode == 2 and "Could not find chrom chrZ" in r.stderr, r.stderr


def test_corrupt_bam_record_ends_the_run_with_a_message_not_a_crash(tmp_path):
    """A record whose l_seq lies about its block (truncated / corrupt BAM): the decoder checks every file-supplied length against
    the block, and the error -- raised lazily on a collector thread in indexed mode -- reaches main(): the driver's own message
    and exit code instead of an out-of-bounds read or std::terminate."""
    exe = driver(tmp_path)
    write_fixture(str(tmp_path))
    sam_to_bam(str(tmp_path / "normal.sam"), str(tmp_path / "normal.bam"), with_index=True, block_bytes=6000)
    sam_to_bam(str(tmp_path / "tumor.sam"), str(tmp_path / "tumor.bam"), with_index=True, block_bytes=6000, corrupt_record=40)
    r = subprocess.run([exe, "--reference", str(tmp_path / "ref.fa"), "--normal", str(tmp_path / "normal.bam"), "--tumor",
                        str(tmp_path / "tumor.bam"), "--region", "chr1:1-6000", "--extract-only", "--extract-threads", "3"],
                       capture_output=True, text=True)
    assert r.returncode == 5, (r.returncode, r.stderr[-400:])
    assert "corrupt BAM record" in r.stderr


def test_sam_record_with_qual_longer_than_seq_is_rejected(tmp_path):
    """CollectFlat sizes the window's quality array by SEQ and copied QUAL at its own length -- a SAM record whose
    QUAL is longer than SEQ (or SEQ '*' with a QUAL) wrote past the array.  The SAM parser now rejects such a record the way the
    BAM decoder rejects a lying length field: message + exit code, no out-of-bounds write."""
    exe = driver(tmp_path)
    write_fixture(str(tmp_path))
    lines = open(tmp_path / "tumor.sam").read().split("\n")
    k = [i for i, ln in enumerate(lines) if ln and ln[0] != "@"][40]
    f = lines[k].split("\t")
    f[10] = f[10] + "I" * 5000
    lines[k] = "\t".join(f)
    (tmp_path / "tumor_bad.sam").write_text("\n".join(lines))
    r = subprocess.run([exe, "--reference", str(tmp_path / "ref.fa"), "--normal", str(tmp_path / "normal.sam"), "--tumor",
                        str(tmp_path / "tumor_bad.sam"), "--region", "chr1:1-6000", "--extract-only", "--extract-threads", "3"],
                       capture_output=True, text=True)
    assert r.returncode == 5, (r.returncode, r.stderr[-400:])
    assert "corrupt SAM record" in r.stderr
