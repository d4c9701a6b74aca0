"""Developer tool: the extract stage of examples/pipeline_driver.cpp on bench.py's synthetic genome, data kept in /tmp/ma_extract.
usage: python tools/dbg/extract_bench.py [threads ...]"""
import os, subprocess, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = "/tmp/ma_extract"
os.makedirs(d, exist_ok=True)
exe = os.path.join(d, "pipeline_driver")
lib = os.path.join(R, "lancet2_amd")
subprocess.check_call(["g++", "-std=c++17", "-O2", "-g", os.path.join(R, "examples", "pipeline_driver.cpp"), "-I", os.path.join(R, "include"),
                       "-L", lib, "-lmicroasm", f"-Wl,-rpath,{lib}", "-Wl,--allow-shlib-undefined", "-DLANCET2_AMD_WITH_ZLIB", "-lz", "-lpthread", "-o", exe])
genome_len, depths, seed = 600_000, (30, 60), 0x5EED
if not os.path.exists(os.path.join(d, "tumor.sam")):
    rng = np.random.default_rng(seed)
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, genome_len)]
    open(os.path.join(d, "ref.fa"), "w").write(">chr1\n" + bytes(genome).decode() + "\n")
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
for nt in [int(x) for x in sys.argv[1:]] or [1, 8]:
    t0 = time.perf_counter()
    r = subprocess.run([exe, "--reference", os.path.join(d, "ref.fa"), "--normal", os.path.join(d, "normal.sam"), "--tumor", os.path.join(d, "tumor.sam"),
                        "--no-active-region", "--extract-only", "--extract-threads", str(nt), "--dump", os.path.join(d, f"dump_{nt}.bin")], capture_output=True, text=True)
    print(nt, "threads:", round(time.perf_counter() - t0, 2), "s;", [l for l in r.stderr.splitlines() if "extract" in l or "windows" in l][-2:])
