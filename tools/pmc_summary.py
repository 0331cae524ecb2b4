#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: per kernel, the mean of every counter over its dispatches."""
import csv
import re
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(<[^>]*>)?)", row["Kernel_Name"])
        k = m.group(1) if m else row["Kernel_Name"][:60]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print("KERNEL", k)
    for c in sorted(d):
        v = d[c]
        print(f"   {c:40s} mean {sum(v)/len(v):16.1f}  n={len(v)}")

# HBM bytes per launch from the fabric-side counters: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies
# the 128-byte requests of wide streaming reads at 64 bytes, hence the factor 2 on the read side.
if len(sys.argv) > 2:
    import json
    out = {}
    for k, d in acc.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"])
            w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
            out[k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes": (2.0 * f + w) * 1024.0, "launches": len(d["FETCH_SIZE"])}
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from polyfem_amd.buildinfo import source_sha256
    # which build the counters belong to: bench.py quotes them only while the sources still hash to this
    out["_meta"] = {"source_sha256": source_sha256()}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
