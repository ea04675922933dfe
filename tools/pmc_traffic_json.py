#!/usr/bin/env python3
"""usage: python3 tools/pmc_traffic_json.py <tag> [round]   (run in the container after `gpurun tools/prof.sh <tag>`)
Builds profiles/<round>_pmc_hbm_traffic.json and profiles/<round>_pmc_ba_traffic.json from the two rocprofv3 --pmc
passes of tools/prof.sh (FETCH_SIZE and WRITE_SIZE collected separately; KB per dispatch, mean per (kernel, grid)).
Both files record the sha of the kernel sources they were taken with (bench.py kernel_source_sha): bench.py
reports a traffic figure only when that sha is the one it is running."""
import collections, csv, glob, json, os, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402

out = collections.defaultdict(dict)
tag = sys.argv[1] if len(sys.argv) > 1 else "r03a"          # gpurun_out/<tag>_fetch, <tag>_write (tools/prof.sh <tag>)
rnd = sys.argv[2] if len(sys.argv) > 2 else "r03"           # profiles/<rnd>_pmc_*.json
for counter, d in (("FETCH_SIZE", tag + "_fetch"), ("WRITE_SIZE", tag + "_write")):
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
head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
meta = {"kernel_source_sha": bench.kernel_source_sha(), "kernel_source_sha_match": bench.kernel_source_sha("match"),
        "kernel_source_sha_ba": bench.kernel_source_sha("ba"), "head": head,
        "command": "tools/prof.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python3 bench.py --steps 1 --warmup 1 "
                   "--cpu-pairs 0 --ba-solves 1 --lines none (separate passes)",
        "units": "KB as reported; HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH correction, MI355X_MICROARCH.md)"}
dst = os.path.join(root, "profiles", f"{rnd}_pmc_hbm_traffic.json")
with open(dst, "w") as f:
    json.dump({"__meta__": meta, **{k: out[k] for k in sorted(out)}}, f, indent=1)
print(dst, len(out), "kernel/grid entries")

# ---- bundle adjustment: HBM bytes of one LM inner iteration (= one tryLambda) of the S200 window ----------
ba = {k: v for k, v in out.items() if k.startswith("eacham::ba_") or k.startswith("eacham::sp_")}
# a tryLambda() starts with ba_schur_groups (landmark groups, round 5) or ba_eliminate_landmarks (pair lists)
tries = sum(v.get("dispatches_FETCH_SIZE", 0) for k, v in ba.items() if (k.startswith("eacham::ba_eliminate") and "grid" in k) or k.startswith("eacham::ba_schur_groups"))
per_kernel, total = {}, 0.0
for k, v in sorted(ba.items()):
    n = v.get("dispatches_FETCH_SIZE", 0)
    b = (2.0 * v.get("FETCH_SIZE_KB_mean_per_dispatch", 0.0) + v.get("WRITE_SIZE_KB_mean_per_dispatch", 0.0)) * 1024.0 * n
    per_kernel[k] = {"dispatches": n, "hbm_bytes_total": b, "hbm_bytes_per_inner_iteration": b / max(tries, 1)}
    total += b
# (no memset belongs to a try: ba_eliminate_landmarks clears the tiles of S itself; the fill kernels of the profiled run are
# the matcher's workspace clears and PyTorch's allocations)
dst = os.path.join(root, "profiles", f"{rnd}_pmc_ba_traffic.json")
with open(dst, "w") as f:
    json.dump({"__meta__": meta, "tries": tries,
               "per_inner_iteration": {"hbm_bytes": total / max(tries, 1),
                                       "note": "sum over the BA kernels (ba_*, sp_*) of (2 FETCH + WRITE) x dispatches / "
                                               "tryLambda calls; the linearisation kernels run once per OUTER iteration and are included "
                                               "(outer = inner on this window)"},
               "kernels": per_kernel}, f, indent=1)
print(dst, "tries", tries, "bytes per inner iteration", total / max(tries, 1))
