#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: per kernel, the mean of every counter over its dispatches."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-60:]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print("KERNEL", k)
    for c in sorted(d):
        v = d[c]
        print(f"   {c:40s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
