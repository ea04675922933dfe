import os, sys, time
sys.path.insert(0, os.getcwd())
from eacham_amd import HipContext, ba, synth
import numpy as np
for which,(nc,nl) in (("s200",(200,50000)),("c4",(500,100000))):
    A = ba.BaArrays.from_scene(synth.make_scene(nc, nl, 10, seed=12345))
    ctx = HipContext(0)
    cfg = ba.OptimizerConfig.refine_ba()
    for _ in range(3): ba.RefineBA(ctx, A, cfg, trace_cap=0)
    ts=[]
    for _ in range(12):
        t0=time.perf_counter(); ba.RefineBA(ctx, A, cfg, trace_cap=0); ts.append(time.perf_counter()-t0)
    print(which, "refine_ba whole call ms: median", round(1e3*float(np.median(ts)),3), "min", round(1e3*min(ts),3))
    ctx.close()
