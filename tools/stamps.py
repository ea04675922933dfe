"""Diagnostic: per-segment wave time of match_tile_kernel (needs the EXP_STAMPS build)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from eacham_amd import HipContext, synth, capi
sc = synth.make_scene(24, 6000, 10)
descs, _ = synth.make_frame_descriptors(sc, 2000, 256)
ctx = HipContext(0)
for f, d in enumerate(descs): ctx.upload_descriptors(f, d)
pairs = synth.all_pairs(24)
L = capi.lib()
buf = (C.c_ulonglong * 16)()
ctx.match_all_pairs(pairs); L.eacham_debug_read(buf, 16, 1)
ctx.match_all_pairs(pairs); L.eacham_debug_read(buf, 16, 1)
v = np.array(list(buf), dtype=np.float64)
n, tiles = v[6], v[7]
names = ["top: prefetch issue", "phase A (8 MFMA + 16 elems)", "phase B", "cm merge+store", "ds_write stage", "barrier"]
print("waves sampled", n, "tiles", tiles)
for i in range(6): print(f"{names[i]:32s} {v[i]/tiles:8.1f} ticks/tile")
print("total", v[:6].sum()/tiles)
