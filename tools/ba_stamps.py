"""Diagnostic: where a Cholesky block step spends its time (needs a library built with -DEXP_BA_STAMPS)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from eacham_amd import HipContext, synth, capi, ba
sc = synth.make_scene(200, 50000, 10)
A = ba.BaArrays.from_scene(sc)
ctx = HipContext(0)
L = capi.lib()
buf = (C.c_ulonglong * 16)()
P = ba.PreparedBA(ctx, A)
cfg = ba.OptimizerConfig.refine_ba()
P.run(cfg); L.eacham_ba_debug_read(buf, 16)
P.run(cfg); L.eacham_ba_debug_read(buf, 16)
v = np.array(list(buf), dtype=np.float64)
names = ["loads + sync", "panel product + sync", "32x32 update + sync", "(unused)", "factor_32"]
print("steps", v[8])
for i in range(5): print(f"{names[i]:24s} {v[i] / v[8]:9.0f} cycles/step")
print("sum", v[:5].sum() / v[8])
print(f"factor wave done at {v[6] / v[8]:.0f}, inverting wave done at {v[5] / v[8]:.0f} cycles after kernel entry")
