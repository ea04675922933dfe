#!/usr/bin/env python3
"""Builds profiles/r01_pmc_hbm_traffic.json from the two rocprofv3 --pmc passes of tools/prof.sh
(FETCH_SIZE and WRITE_SIZE collected separately; KB per dispatch, mean per (kernel, grid))."""
import collections, csv, glob, json, os, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = collections.defaultdict(dict)
for counter, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    acc = collections.defaultdict(list)
    for fn in glob.glob(os.path.join(root, "gpurun_out", d, "**", "*counter_collection.csv"), recursive=True):
        with open(fn) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[f"{name} grid={row['Grid_Size']}"].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        out[k][f"{counter}_KB_mean_per_dispatch"] = sum(v) / len(v)
        out[k][f"dispatches_{counter}"] = len(v)
dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "profiles", "r01_pmc_hbm_traffic.json")
with open(dst, "w") as f:
    json.dump({k: out[k] for k in sorted(out)}, f, indent=1)
print(dst, len(out), "kernel/grid entries")
