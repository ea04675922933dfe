"""Diagnostic: per-stage device time of one BA inner iteration (event timers of the library)."""
import os, sys
sys.path.insert(0, os.getcwd())
from eacham_amd import HipContext, synth, ba, capi
sc = synth.make_scene(200, 50000, 10)
A = ba.BaArrays.from_scene(sc)
ctx = HipContext(0)
s = ba.PreparedBA(ctx, A)
cfg = ba.OptimizerConfig.refine_ba()
s.run(cfg)
ctx.profile_reset(); ctx.profile_enable(True)
n = 0
for _ in range(5): n += s.run(cfg).inner_iterations
ctx.profile_enable(False)
for name, kid in [("linearize", capi.KERNEL_BA_LINEARIZE), ("schur", capi.KERNEL_BA_SCHUR), ("solve", capi.KERNEL_BA_SOLVE), ("error", capi.KERNEL_BA_ERROR)]:
    print(f"{name:10s} {ctx.profile_get(kid)[1] / n * 1e3:8.1f} us per inner iteration")
s.close()
