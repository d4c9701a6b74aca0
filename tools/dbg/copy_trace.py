"""Developer tool: summary of a rocprofv3 --memory-copy-trace / --kernel-trace csv directory (big copies, copy kernels)."""
import csv, glob, sys
d = sys.argv[1]
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), list(rows[0].keys()) if rows else None)
    tot = {}
    for r in rows:
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        t = tot.setdefault(r.get("Direction", "?"), [0, 0])
        t[0] += 1; t[1] += dur
    print({k: (c, round(ns / 1e6, 2)) for k, (c, ns) in tot.items()})
    big = sorted(rows, key=lambda r: int(r["Start_Timestamp"]) - int(r["End_Timestamp"]))[:12]
    for r in big:
        print("  ", r.get("Direction"), round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3), "ms", {k: v for k, v in r.items() if "ize" in k or "ytes" in k or "gent" in k})
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    agg = {}
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"][:60]
        if "rocclr" in nm or "copy" in nm.lower() or "fill" in nm.lower():
            a = agg.setdefault(nm, [0, 0, 0])
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            a[0] += 1; a[1] += dur; a[2] = max(a[2], dur)
    for nm, (c, ns, mx) in agg.items():
        print(nm, "calls", c, "total ms", round(ns / 1e6, 2), "max ms", round(mx / 1e6, 3))
