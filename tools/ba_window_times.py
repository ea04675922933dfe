"""Where a local-window RefineBA call spends its time: prepare (structure build + allocation + upload) / LM / release.
python3 tools/ba_window_times.py   (TUM stand-in windows of bench.py's c3_tum line)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from eacham_amd import HipContext, synth, ba
tum = synth.make_scene(500, 30_000, 10, seed=3)
wins = [ba.BaArrays.from_scene(synth.local_window(tum, f)) for f in range(100, 140)]
ctx = HipContext(0)
n = len(wins)
cfg = ba.OptimizerConfig.refine_ba()
ba.RefineBA(ctx, wins[0], cfg)
tp = tr = tc = 0.0
inner = 0
pu = np.zeros(3)
for A in wins:
    t0 = time.perf_counter(); P = ba.PreparedBA(ctx, A); ctx.sync()
    pu += P.plan_info()["prepare_us"]
    t1 = time.perf_counter(); o = P.run(cfg, trace_cap=0)
    t2 = time.perf_counter(); P.close()
    t3 = time.perf_counter()
    tp += t1 - t0; tr += t2 - t1; tc += t3 - t2; inner += o.inner_iterations
print(f"inside eacham_ba_prepare: structures {pu[0]/n:.0f} us, plan {pu[1]/n:.0f} us, arena + upload + sync {pu[2]/n:.0f} us")
print(f"per window: prepare {1e3*tp/n:.3f} ms, LM {1e3*tr/n:.3f} ms ({inner/n:.1f} inner iterations, {1e3*tr/inner:.3f} ms each), release {1e3*tc/n:.3f} ms")
t0 = time.perf_counter()
for A in wins: ba.RefineBA(ctx, A, cfg, trace_cap=0)
print(f"RefineBA end to end: {1e3*(time.perf_counter()-t0)/n:.3f} ms per window")
