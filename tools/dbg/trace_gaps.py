"""Developer tool: from a rocprofv3 --kernel-trace (+ --memory-copy-trace) csv directory, the GPU's busy / idle timeline:
union of kernel intervals, gaps > 0.2 ms with what runs either side, copies with size and rate."""
import csv, glob, sys
d = sys.argv[1]
ks = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
cs = []
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "?"), int(r.get("Size", 0) or 0)))
ks.sort()
t0 = ks[0][0]
tail = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0  # analyse the last `tail` ms
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.2       # report gaps longer than this (ms)
tend = max(e for _, e, _ in ks)
lo = tend - int(tail * 1e6)
cur_s, cur_e = None, None
busy = 0
gaps = []
last_name = ""
for s, e, nm in ks:
    if e < lo:
        continue
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s > cur_e:
        busy += cur_e - cur_s
        gaps.append((cur_e, s, last_name, nm))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    last_name = nm
busy += cur_e - cur_s
print("window %.1f ms, kernels busy (union) %.1f ms, idle %.1f ms" % ((tend - lo) / 1e6, busy / 1e6, (tend - lo - busy) / 1e6))
import collections
agg = collections.Counter()
for a, b, n1, n2 in gaps:
    agg[(n1[:28], n2[:28])] += b - a
for (n1, n2), v in agg.most_common(25):
    print("  idle %.2f ms in all between %s -> %s" % (v / 1e6, n1, n2))
for a, b, n1, n2 in gaps:
    if b - a > thr * 1e6:
        print("  gap %.2f ms at t=%.1f ms: after %s, before %s" % ((b - a) / 1e6, (a - lo) / 1e6, n1, n2))
big = [(s, e, dr, sz) for s, e, dr, sz in cs if e >= lo and sz > (1 << 20)]
tot = {}
for s, e, dr, sz in big:
    k = dr
    t = tot.setdefault(k, [0, 0, 0])
    t[0] += 1; t[1] += sz; t[2] += e - s
for k, (c, sz, ns) in tot.items():
    print("  copies %s: %d, %.1f MB, %.1f ms busy, %.1f GB/s" % (k, c, sz / 1e6, ns / 1e6, sz / max(ns, 1)))
# concurrency: mean number of kernels running
ev = []
for s, e, _ in ks:
    if e >= lo:
        ev.append((max(s, lo), 1)); ev.append((e, -1))
ev.sort()
acc = 0; cur = 0; prev = lo
for t, dlt in ev:
    acc += cur * (t - prev); prev = t; cur += dlt
print("mean kernels in flight: %.2f" % (acc / (tend - lo)))
