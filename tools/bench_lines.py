"""Prints one short row per JSON line of a bench.py log: python3 tools/bench_lines.py <log>"""
import json, sys
for l in open(sys.argv[1]):
    if not l.startswith("{"):
        continue
    d = json.loads(l)
    rf = d.get("roofline", {})
    extra = ""
    if "ba" in d and isinstance(d["ba"], dict) and "windows_per_s" in d["ba"]:
        extra = f" windows/s {d['ba']['windows_per_s']:.0f}"
    if "ms_per_inner_iter" in d:
        extra += f" ms/inner {d['ms_per_inner_iter']:.4f}"
    print(f"{d.get('line', 'HEAD'):16s} {d['value']:14.1f} {d.get('unit', ''):18s} frac {rf.get('frac', float('nan')):.4f}{extra}")
