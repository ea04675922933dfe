"""Diagnostic (-DEXP_BA_STAMPS build): start / end of every workgroup of the first Cholesky launch of config 4."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from eacham_amd import HipContext, synth, capi, ba
A = ba.BaArrays.from_scene(synth.make_scene(500, 100000, 10, seed=4))
ctx = HipContext(0)
L = capi.lib()
P = ba.PreparedBA(ctx, A)
cfg = ba.OptimizerConfig.refine_ba()
P.run(cfg); P.run(cfg)
buf = (C.c_ulonglong * 3600)()
L.eacham_ba_debug_wg_times(buf, 3600)
v = np.array(list(buf), dtype=np.int64).reshape(1200, 3)[:1128]
t0 = v[1:, 0].min()
st = (v[:, 0] - t0) / 100.0  # us (100 MHz)
en = (v[:, 1] - t0) / 100.0
en[0] = np.nan
hw = v[:, 2]
print("start us: min %.1f  p50 %.1f  p90 %.1f  max %.1f" % (st.min(), np.median(st), np.percentile(st, 90), st.max()))
d = en[1:] - st[1:]
print("duration us (bulk tiles): min %.1f p50 %.1f p90 %.1f max %.1f" % (d.min(), np.median(d), np.percentile(d, 90), d.max()))
print("last end %.1f us" % np.nanmax(en))
for lo in range(0, 1128, 128):
    sl = slice(max(lo, 1), lo + 128)
    print("blocks %4d..: start %.1f..%.1f  dur p50 %.1f" % (lo, st[sl].min(), st[sl].max(), np.median(en[sl] - st[sl])))
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
print("tile 0: HW_ID %x" % hw[0], "; workgroups with tile 0's HW_ID cu/sh/se and XCD:", np.sum(((hw >> 8) & 0xff) == ((hw[0] >> 8) & 0xff)))
# concurrency: how many workgroups are running at time t
for t in (2, 6, 10, 14, 20, 30, 40):
    print("t=%2d us: running %d" % (t, int(np.sum((st[1:] <= t) & (en[1:] > t)))))
