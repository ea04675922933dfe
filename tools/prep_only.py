"""Five warm eacham_ba_prepare calls (device form) of S200 / config 4 and nothing else: python3 tools/prep_only.py [s200|c4]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
from eacham_amd import HipContext, ba, synth
which = sys.argv[1] if len(sys.argv) > 1 else "s200"
nc, nl = (200, 50_000) if which == "s200" else (500, 100_000)
A = ba.BaArrays.from_scene(synth.make_scene(nc, nl, 10, seed=12345))
ctx = HipContext(0)
for it in range(5):
    t0 = time.perf_counter()
    pb = ba.PreparedBA(ctx, A)
    t1 = time.perf_counter()
    print(which, "prepare wall ms", round(1e3 * (t1 - t0), 3), "parts us", pb.plan_info()["prepare_us"])
    pb.close()
ctx.close()
