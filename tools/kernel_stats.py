"""Per-kernel table of a rocprofv3 --kernel-trace --stats run: python3 tools/kernel_stats.py <dir> [divide-by]"""
import csv, glob, sys
d = sys.argv[1]
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
fn = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[-1]
tot = 0.0
for r in csv.DictReader(open(fn)):
    name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("eacham::", "")[:52]
    t = float(r["TotalDurationNs"]) / 1e3
    tot += t
    print(f"{name:54s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:9.2f} us  per-unit {t/div:9.2f} us")
print(f"total {tot/div:.1f} us per unit")
