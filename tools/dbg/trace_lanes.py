"""Developer tool: per-queue view of a rocprofv3 --kernel-trace csv directory over the last `tail` ms: busy time, idle time,
and where the idle time sits (which kernel ran before / after each gap, summed by that pair)."""
import csv, glob, sys, collections
d = sys.argv[1]
tail = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-28:], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
tend = max(r[1] for r in rows)
lo = tend - int(tail * 1e6)
byq = collections.defaultdict(list)
for s, e, n, q, st in rows:
    if e >= lo:
        byq[(q, st)].append((s, e, n))
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    ks.sort()
    busy = sum(e - s for s, e, _ in ks)
    span = ks[-1][1] - ks[0][0]
    gaps = collections.Counter()
    for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
        if s1 - e0 > 20_000:
            gaps[(n0, n1)] += s1 - e0
    print("queue/stream %s: %d kernels, busy %.1f ms of %.1f ms" % (q, len(ks), busy / 1e6, span / 1e6))
    for (a, b), ns in gaps.most_common(14):
        print("     idle %.2f ms between %s -> %s" % (ns / 1e6, a, b))
