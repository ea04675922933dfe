#!/usr/bin/env python3
"""usage: tools/pmc_sq_json.py [tiles] [round] [prefix] [kernel] [suffix] [dim]
Builds profiles/<round>_pmc_sq_<kernel>.json (kernel defaults to match_sweep_kernel) from the three SQ counter passes of tools/pmc.sh
(gpurun_out/<prefix>_a, _b, _c; defaults r01, sq): mean per dispatch of the LARGEST-grid launch of that kernel + the
derived fractions quoted in DESIGN.md (formulas in the `derived_from` field)."""
import collections, csv, glob, json, os, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
prefix = sys.argv[3] if len(sys.argv) > 3 else "sq"
kernel = sys.argv[4] if len(sys.argv) > 4 else "match_sweep_kernel"
suffix = sys.argv[5] if len(sys.argv) > 5 else ""      # e.g. _d128: profiles/<round>_pmc_sq_<kernel>_d128.json
dim = sys.argv[6] if len(sys.argv) > 6 else "256"
for d in (prefix + "_a", prefix + "_b", prefix + "_c"):
    for fn in glob.glob(os.path.join(root, "gpurun_out", d, "**", "*counter_collection.csv"), recursive=True):
        with open(fn) as f:
            for row in csv.DictReader(f):
                if kernel in row["Kernel_Name"]:
                    acc[int(row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
grid = max(acc)
c = {k: sum(v) / len(v) for k, v in acc[grid].items()}
tiles = float(sys.argv[1]) if len(sys.argv) > 1 else 64.0
cyc = c["GRBM_GUI_ACTIVE"] / 8.0
simds = 256 * 4
out = {
    "kernel": f"{kernel} ({dim}-D), grid {grid} threads, 64-frame reduced S200 set (tools/pmc.sh, three --pmc passes)",
    "counters_mean_per_dispatch": dict(sorted(c.items())),
    "derived": {
        "gpu_cycles_per_dispatch": cyc,
        "valu_busy_frac": c["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * simds),
        "mfma_busy_frac": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * simds),
        "valu_mfma_coexec_frac": c["SQ_VALU_MFMA_COEXEC_CYCLES"] / (cyc * simds),
        "valu_insts_per_wave_tile": c["SQ_INSTS_VALU"] / c["SQ_WAVES"] / tiles,
        "mfma_insts_per_wave_tile": c["SQ_INSTS_MFMA"] / c["SQ_WAVES"] / tiles,
        "waves_resident_per_simd": c["SQ_WAVE_CYCLES"] * 4 / (cyc * simds),
        "wave_time_split": {"issuing": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                            "waiting": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                            "issue_stalled": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]},
    },
    "extra": {k: c[k] / c["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS") if k in c and "SQ_WAVE_CYCLES" in c},
    "derived_from": "cycles = GRBM_GUI_ACTIVE/8 (sum over 8 XCDs); SQ_ACTIVE_*/SQ_WAVE_CYCLES/SQ_WAIT_* count quad-cycles; "
                    "busy fractions are per SIMD (1024 SIMDs); per-wave-tile counts divide by SQ_WAVES and the tiles of a sweep",
}
dst = os.path.join(root, "profiles", f"{rnd}_pmc_sq_{kernel}{suffix}.json")
with open(dst, "w") as f:
    json.dump(out, f, indent=1)
print(dst, json.dumps(out["derived"]))
