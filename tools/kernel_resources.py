"""Developer tool: per-kernel resource table of the gfx950 code objects (VGPRs, AGPRs, spills, scratch, static LDS, the
occupancy the compiler derives) from hipcc's -Rpass-analysis=kernel-resource-usage, one row per kernel / instantiation.
Dynamic LDS (k_clean_chains, k_insert, k_align_reg*, ...) is set at launch and is not in the code object: see the launch
sites.  usage: python tools/kernel_resources.py > profiles/r4_kernel_resources.txt   (no GPU needed)"""
import glob
import os
import re
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for src in sorted(glob.glob(os.path.join(R, "lancet2_amd", "csrc", "*.hip"))):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-function",
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    for line in out.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(anonymous namespace\)::", "", name)
            name = re.sub(r"^void ", "", name).replace("ma::", "")
            name = re.sub(r"\((GArgs|MsaArgs|CleanArgs|ChainArgs|CxArgs|DBatch|GraphWs|ma_asm_out|unsigned|RegClass2|KeyBase)[^)]*\)$", "", name)
            cur = {"file": os.path.basename(src), "kernel": name[:72]}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
print("%-12s %-72s %5s %5s %7s %7s %8s %9s %4s" % ("file", "kernel", "VGPR", "AGPR", "v-spill", "s-spill", "scratch", "LDS(stat)", "occ"))
for r in rows:
    print("%-12s %-72s %5d %5d %7d %7d %8d %9d %4d" % (r["file"], r["kernel"], r.get("vgpr", 0), r.get("agpr", 0), r.get("vspill", 0),
                                                       r.get("sspill", 0), r.get("scratch", 0), r.get("lds", 0), r.get("occ", 0)))
