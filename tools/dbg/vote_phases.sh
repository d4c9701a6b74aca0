# Developer tool: where k_vote's time goes, by differencing timing builds (MA_VOTE_STOP=1: set-up only, =2: the hint trips only, =3: hint trips + group vote).
# Build the variants first (see the hipcc lines in the round-5 notes of DESIGN.md section 9); results of the variants are INVALID.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
export MA_BENCH_CACHE=/tmp/mbc
python3 bench.py --no-cpu --no-also --gen-only > /dev/null 2>&1
for lib in ${VOTE_LIBS:-libmicroasm.so libmicroasm_vs1.so libmicroasm_vs2.so libmicroasm_vs3.so}; do
  MA_LIB=$PWD/lancet2_amd/$lib MA_STREAMS=1 python3 - >> gpurun_out/r5b/vote_phases.txt 2>/dev/null <<PY
import sys, json, subprocess, os
r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--no-cpu", "--no-also"], capture_output=True, text=True)
line = [l for l in r.stdout.splitlines() if l.startswith("{")]
if line:
    d = json.loads(line[-1]); print("$lib", "k_vote", d["kernel_ms_per_step"].get("k_vote"), "ms per 16384 windows single-lane; parity_sample", d.get("parity_sample"))
else:
    print("$lib", "no line; rc", r.returncode, r.stderr[-300:])
PY
done
