"""Warm timing of eacham_ba_prepare (its three parts) and of a whole RefineBA call, host-built vs device-built structure:
python3 tools/ba_prepare_times.py [s200|c4]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from eacham_amd import HipContext, ba, synth
which = sys.argv[1] if len(sys.argv) > 1 else "s200"
nc, nl = (200, 50_000) if which == "s200" else (500, 100_000)
A = ba.BaArrays.from_scene(synth.make_scene(nc, nl, 10, seed=12345))
for mode in ["host", "device"]:
    os.environ["EACHAM_BA_PREPARE"] = mode
    ctx = HipContext(0)
    rows = []
    for it in range(6):
        t0 = time.perf_counter()
        pb = ba.PreparedBA(ctx, A)
        t1 = time.perf_counter()
        rows.append((1e3 * (t1 - t0), pb.plan_info()["prepare_us"]))
        pb.close()
    whole = []
    for it in range(6):
        t0 = time.perf_counter()
        out = ba.RefineBA(ctx, A, ba.OptimizerConfig.global_ba())
        whole.append(1e3 * (time.perf_counter() - t0))
    print(which, mode, "prepare wall ms", [round(r[0], 2) for r in rows], "parts us", rows[-1][1], "RefineBA(-1) wall ms", [round(w, 2) for w in whole],
          "iters", out.outer_iterations, out.inner_iterations)
    ctx.close()
