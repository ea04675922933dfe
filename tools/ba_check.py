"""S200 RefineBA against the oracle: python3 tools/ba_check.py  (EACHAM_HIP_LIB / EACHAM_BA_PREPARE / EACHAM_BA_SCHUR select the build and the form)"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from eacham_amd import HipContext, synth, ba
import oracle_api as O
cams = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lms = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
A = ba.BaArrays.from_scene(synth.make_scene(cams, lms, 10))
ctx = HipContext(0)
cfg = ba.OptimizerConfig.refine_ba()
out = ba.RefineBA(ctx, A, cfg)
ref = O.ba_solve(A, cfg)
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
print("lib", os.environ.get("EACHAM_HIP_LIB", "product"), "iters", out.outer_iterations, out.inner_iterations, "ref", ref.outer_iterations, ref.inner_iterations,
      "rel poses %.2e points %.2e" % (rel(out.cam_T_wc, ref.cam_T_wc), rel(out.points, ref.points)), "err", out.final_error, ref.final_error)
s = ba.PreparedBA(ctx, A)
try:
    gg = s.structure("g_groups").reshape(-1, 8)
    if len(gg):
        print("groups", len(gg), "segments", int(gg[:, 7].sum()), "entries", int(gg[:, 6].sum()), "ent padded", s.structure("g_ent").size,
              "chunks", s.structure("g_chunks").size // 2, "blocks", s.structure("g_blk").size // 4, "long", s.structure("g_longblk").size)
finally:
    s.close()
