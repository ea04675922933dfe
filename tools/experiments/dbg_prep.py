import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from eacham_amd import HipContext, ba, synth
os.environ["EACHAM_BA_PREPARE"] = "host"; host = HipContext(0)
os.environ["EACHAM_BA_PREPARE"] = "device"; dev = HipContext(0)
A = ba.BaArrays.from_scene(synth.make_scene(5, 80, 3, seed=12, pixel_noise=1.5))
a, b = ba.PreparedBA(host, A), ba.PreparedBA(dev, A)
for name in ba.PreparedBA.STRUCTURE:
    x, y = a.structure(name), b.structure(name)
    if x.shape != y.shape or not np.array_equal(x, y):
        idx = np.nonzero(x != y)[0] if x.shape == y.shape else []
        print(name, "DIFF", x.shape, y.shape, idx[:10], [(x[i].hex() if x.dtype == np.float64 else x[i], y[i].hex() if y.dtype == np.float64 else y[i]) for i in idx[:6]])
    else:
        print(name, "ok")
print(A.cam_T_wc[1])
