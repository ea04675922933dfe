#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per dispatch of a kernel."""
import csv, sys, glob, collections
pat = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "match_tile_kernel"
for fn in sorted(glob.glob(pat, recursive=True)):
    acc = collections.defaultdict(list)
    with open(fn) as f:
        for row in csv.DictReader(f):
            if kern in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
