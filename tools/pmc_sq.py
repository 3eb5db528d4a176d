#!/usr/bin/env python3
"""Mean SQ/TA/TCC counter values per launch of each kernel from rocprofv3 --pmc output directories."""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            a = acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    if not ("k_dwp" in k or "k_fwd<0" in k or "k_dx" in k or "k_fwd64" in k):
        continue
    print(k)
    for c in sorted(acc[k]):
        s, n = acc[k][c]
        print("   %-36s %16.1f  (n=%d)" % (c, s / n, n))
