"""Timeline of the kernels of ONE local-window RefineBA call from a rocprofv3 --kernel-trace run of tools/ba_window_times.py:
python3 tools/win_timeline.py <dir> [call index from the end, default 3]   — start (us), duration, gap to the previous kernel's end."""
import csv, glob, sys
fn = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r["Start_Timestamp"]))
# a call starts with its initial error pass: ba_error_landmarks / ba_retract_cameras with apply_delta = 0 come once per call
starts = [i for i, r in enumerate(rows) if "ba_error_landmarks" in r["Kernel_Name"]]
lo, hi = starts[-back], starts[-back + 1]
rows = rows[lo - 1:hi - 1]
t0 = int(rows[0]["Start_Timestamp"])
prev = t0
busy = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("eacham::", "")[:40]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:6.1f}  gap {(s - prev) / 1e3:6.1f}  {name}")
    busy += e - s
    prev = max(prev, e)
print(f"span {(prev - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, {len(rows)} launches")
