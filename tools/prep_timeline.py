"""Timeline of ONE warm eacham_ba_prepare from a rocprofv3 --kernel-trace run of tools/prep_only.py:
python3 tools/prep_timeline.py <dir>   — kernels of the last prepare call in start order: start (us from the first), duration, gap before."""
import csv, glob, sys
fn = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r["Start_Timestamp"]))
# the last prepare call = the kernels behind the last prep_values launch
last = max(i for i, r in enumerate(rows) if "prep_keys_lm" in r["Kernel_Name"])
rows = rows[last:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("eacham::", "")[:44]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  {name}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, {len(rows)} launches")
