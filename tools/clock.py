"""Diagnostic: in-kernel clock and wave cycles per tile of match_tile_kernel under sustained load
(needs a library built with -DEXP_CLOCK; MI355X_MICROARCH.md 'DVFS give-back' item 6).
usage: EACHAM_HIP_LIB=ab_libs/x_clock.so python tools/clock.py"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from eacham_amd import HipContext, synth, capi
sc = synth.make_scene(64, 16000, 10)
descs, _ = synth.make_frame_descriptors(sc, 2000, 256)
ctx = HipContext(0)
for f, d in enumerate(descs): ctx.upload_descriptors(f, d)
pairs = synth.all_pairs(64)[:1800]          # one full launch
L = capi.lib()
buf = (C.c_ulonglong * 16)()
t0 = time.time()
while time.time() - t0 < 2.5: ctx.match_all_pairs(pairs)
L.eacham_debug_read(buf, 16, 1)
t0 = time.time(); n = 0
while time.time() - t0 < 1.0: ctx.match_all_pairs(pairs); n += 1
L.eacham_debug_read(buf, 16, 1)
v = np.array(list(buf), dtype=np.float64)
print(f"launches {n}  waves sampled {v[10]:.0f}  in-kernel clock {v[8] / v[9] * 0.1:.3f} GHz  "
      f"wave cycles per tile {v[8] / v[11]:.0f}  realtime ns per tile {v[9] / v[11] * 10:.0f}")
