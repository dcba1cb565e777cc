#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel from the counter_collection CSVs under a directory."""
import csv, glob, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_lqer_gemm"
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            a = acc[row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(f"{k:32s} {acc[k][0]/acc[k][1]:16.1f}  (n={acc[k][1]})")
