"""Throughput of the float-descriptor matcher path (fp32 MFMA): F frames of 2000 x 256 unit-norm rows."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from eacham_amd import HipContext, synth, capi
F = int(sys.argv[1]) if len(sys.argv) > 1 else 24
base = synth.unit_float_descriptors(2000, 256, 1, 99)
ctx = HipContext(0)
for f in range(F):
    ctx.upload_descriptors_f32(f, synth.unit_float_descriptors(2000, 256, 1, f, shared=base[:1000]))
pairs = synth.all_pairs(F)
ctx.match_all_pairs(pairs)
ctx.profile_reset(); ctx.profile_enable(True)
t0 = time.perf_counter(); c = ctx.match_all_pairs(pairs); dt = time.perf_counter() - t0
n, ms = ctx.profile_get(capi.KERNEL_MATCH_TILE)
print(f"{len(pairs)} pairs: {len(pairs)/dt:.0f} pairs/s end to end (incl. host copies), tile kernel {ms:.2f} ms -> "
      f"{len(pairs)/(ms*1e-3):.0f} pairs/s = {len(pairs)*2*2000*2000*256/(ms*1e-3)/1e12:.1f} TFLOP/s fp32 (peak 157.3); matches {c[0].sum()}")
