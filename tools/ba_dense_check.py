"""Diagnostic: the dense form of the Schur stage on one local window (structure, one damped step, LM run, timing)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from eacham_amd import ba, synth, HipContext
import oracle_api as O

def ctx_with(**env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return HipContext(0)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v

def rel(a, b): return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))

scene = synth.make_scene(60, 4000, 8, seed=5, pixel_noise=1.0)
A = ba.BaArrays.from_scene(synth.local_window(scene, 30, max_neighbours=18))
A.obs_uv[::17] += 25.0
cfg = ba.OptimizerConfig.refine_ba()
ref = O.ba_solve(A, cfg)
So, go, dco, dlo, erro, lino, ok = O.ba_step(A, 1e-3, 0)
for mode in ("pairs", "dense"):
    ctx = ctx_with(EACHAM_BA_SCHUR=mode)
    pb = ba.PreparedBA(ctx, A)
    print(mode, "w_groups", len(pb.structure("w_groups")) // 2, "blocks", len(pb.structure("blocks")) // 4, pb.plan_info())
    pb.close()
    S, g, dc, dl, err, lin = ba.debug_step(ctx, A, 1e-3)
    print(mode, "step: S", rel(S, So), "g", rel(g, go), "dc", rel(dc, dco), "dl", rel(dl, dlo), "err", err, erro, "lin", lin, lino)
    out = ba.RefineBA(ctx, A, cfg)
    print(mode, "LM:", out.outer_iterations, out.inner_iterations, "ref", ref.outer_iterations, ref.inner_iterations,
          "T", rel(out.cam_T_wc, ref.cam_T_wc), "pts", rel(out.points, ref.points), "final", out.final_error, ref.final_error)
    t0 = time.perf_counter()
    for _ in range(50):
        ba.RefineBA(ctx, A, cfg, trace_cap=0)
    print(mode, "ms per window call", (time.perf_counter() - t0) / 50 * 1e3, "inner", out.inner_iterations)
    ctx.close()
