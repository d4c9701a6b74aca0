"""Developer tool: aggregate a rocprofv3 --pmc counter_collection.csv per kernel name."""
import csv
import glob
import re
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            m = re.search(r"(k_[a-z_0-9]+|gate_kernel)", row["Kernel_Name"])
            name = m.group(1) if m else row["Kernel_Name"].split("(")[0][:40]
            agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[(name, row["Dispatch_Id"])] += 0
ndisp = defaultdict(set)
for (name, d) in calls:
    ndisp[name].add(d)
names = sorted(agg)
ctrs = sorted({c for n in names for c in agg[n]})
print("kernel".ljust(28), "disp", *[c[-22:].rjust(22) for c in ctrs])
for n in names:
    print(n[:28].ljust(28), str(len(ndisp[n])).rjust(4), *[("%.4g" % agg[n][c]).rjust(22) for c in ctrs])
