R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5d
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 $R/bench.py --config C4 --windows 2048 --distinct 512 --steps 2 --warmup 1 --no-cpu --no-also --str-every 0 --hard-every 0 --gen-workers 1 > $O/bench.json 2> $O/bench.err
cd $R
python3 tools/dbg/trace_gaps.py $O/trace 1500 > $O/gaps.txt 2>&1
python3 - > $O/ksum.txt <<'PY'
import csv, glob, os
O = os.environ.get("O", "gpurun_out/r5d")
tot = {}
for f in glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0][:50]
        t = tot.setdefault(n, [0, 0])
        t[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); t[1] += 1
for n, (ns, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:40]:
    print("%-52s %9.2f ms %6d launches" % (n, ns / 1e6, c))
PY
rm -rf $O/trace
