"""Runs RefineBA on the S200 (or a scaled) window a few times — target for rocprofv3."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from eacham_amd import HipContext, synth, ba
nc, nl = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 50000)
sc = synth.make_scene(nc, nl, 10)
A = ba.BaArrays.from_scene(sc)
ctx = HipContext(0)
s = ba.PreparedBA(ctx, A)
s.run(ba.OptimizerConfig.refine_ba())
t0 = time.perf_counter(); n = 0
for _ in range(5):
    o = s.run(ba.OptimizerConfig.refine_ba()); n += o.inner_iterations
dt = time.perf_counter() - t0
print(f"{nc} cams {nl} lm: {n} inner iters in {dt*1e3:.1f} ms -> {dt/n*1e3:.3f} ms/iter, final error {o.final_error:.6g}")
s.close()
