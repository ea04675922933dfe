"""Kernel timeline of ONE added frame of the incremental loop from a rocprofv3 --kernel-trace run of the loop driver:
python3 tools/loop_timeline.py <dir> [frame from the end, default 20]  — the frame starts at its first solve_pnp_front_kernel."""
import csv, glob, sys
fn = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r["Start_Timestamp"]))
# an added frame starts with the first EPnP chunk of its solvePnPRansac: a front kernel more than 1 ms after the previous one's
fronts = [i for i, r in enumerate(rows) if "solve_pnp_front" in r["Kernel_Name"]]
starts = [fronts[0]]
for a, b in zip(fronts, fronts[1:]):
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]) > 1_000_000:
        starts.append(b)
lo, hi = starts[-back], starts[-back + 1]
rows = rows[lo:hi]
t0 = int(rows[0]["Start_Timestamp"])
prev = t0
busy = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("eacham::", "")[:40]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:6.1f}  gap {(s - prev) / 1e3:6.1f}  {name}")
    busy += e - s
    prev = max(prev, e)
print(f"span {(prev - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, {len(rows)} launches, {len(starts)} frames found")
