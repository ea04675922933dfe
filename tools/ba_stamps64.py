"""Diagnostic: where tile (0,0) of a 64-column Cholesky step spends its time (library built with -DEXP_BA_STAMPS)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from eacham_amd import HipContext, synth, capi, ba
A = ba.BaArrays.from_scene(synth.make_scene(200, 50000, 10))
ctx = HipContext(0)
L = capi.lib()
buf = (C.c_ulonglong * 16)()
P = ba.PreparedBA(ctx, A)
cfg = ba.OptimizerConfig.refine_ba()
P.run(cfg); L.eacham_ba_debug_read(buf, 16)
P.run(cfg); L.eacham_ba_debug_read(buf, 16)
v = np.array(list(buf), dtype=np.float64)
names = ["loads + panel + sync", "first quadrant + sync", "factor a", "wait: inverter a, rows 32..63", "dump T10/T11 + sync",
         "L_ba + sync", "D_b + sync", "factor b", "", "wait: inverter b, L_ba W_a", "X"]
print("steps", v[15])
for i, nm in enumerate(names):
    if nm: print(f"{nm:32s} {v[i] / v[15]:9.0f} cycles/step")
print("sum", v[:11].sum() / v[15])
print(f"after the first-quadrant barrier: inverter a done at {v[12]/v[15]:.0f}, wave 2 updated at {v[13]/v[15]:.0f}, wave 3 updated at {v[11]/v[15]:.0f}, panel stored at {v[14]/v[15]:.0f}")
