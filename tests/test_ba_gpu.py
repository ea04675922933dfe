"""GPU: the HIP bundle adjuster (through the C-ABI) against the CPU oracle.
Tolerance (north_star): camera poses / landmark positions within 1e-5 relative; the tests ask for
far tighter agreement wherever fp64 + identical algorithm allow it."""
import os

import numpy as np
import pytest

from eacham_amd import ba, synth
from eacham_amd import capi, EachamError
import oracle_api as O

pytestmark = pytest.mark.gpu
GOLD_DIR = os.path.join(os.path.dirname(__file__), "golden")
POSE_POINT_RTOL = 1e-5  # north_star tolerance


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def scene_arrays(seed=3, n_cams=6, n_lm=200, k=4, outliers=True, **kw):
    sc = synth.make_scene(n_cams, n_lm, k, seed=seed, **kw)
    A = ba.BaArrays.from_scene(sc)
    if outliers:
        A.obs_uv[::13] += 30.0
    return sc, A


@pytest.mark.parametrize("lam", [0.0, 1e-4, 10.0])
def test_damped_step_matches_oracle(hip_ctx, lam):
    """K3-K7 in one shot: reduced system, its rhs, both parts of the step, error, linear cost change."""
    sc, A = scene_arrays(seed=9, n_cams=7, n_lm=150, k=4, pixel_noise=2.0)
    S, g, dc, dl, err, lin = ba.debug_step(hip_ctx, A, lam)
    So, go, dco, dlo, erro, lino, ok = O.ba_step(A, lam, 0)
    assert ok
    assert rel(S, So) < 1e-11 and rel(g, go) < 1e-11
    assert rel(dc, dco) < 1e-8 and rel(dl, dlo) < 1e-8
    assert np.isclose(err, erro, rtol=1e-12) and np.isclose(lin, lino, rtol=1e-9)


def test_step_with_repeated_camera_and_cheirality(hip_ctx):
    sc, A = scene_arrays(seed=4, n_cams=5, n_lm=80, k=3)
    # the same camera observes one landmark twice (two keypoints -> one map point)
    A.obs_cam = np.concatenate([A.obs_cam, A.obs_cam[:3]]).astype(np.uint32)
    A.obs_point = np.concatenate([A.obs_point, A.obs_point[:3]]).astype(np.uint32)
    A.obs_uv = np.concatenate([A.obs_uv, A.obs_uv[:3] + 0.7])
    # a landmark behind its cameras: zero residual / Jacobians for those factors
    A.points[5] = A.points[5] + 50.0 * (np.linalg.inv(A.cam_T_wc[int(A.obs_cam[15])])[:3, 3] - A.points[5])
    # and an unobserved landmark
    A.points = np.concatenate([A.points, [[0.1, 0.2, 0.3]]])
    A.point_observers = np.concatenate([A.point_observers, [2]]).astype(np.int32)
    S, g, dc, dl, err, lin = ba.debug_step(hip_ctx, A, 1e-3)
    So, go, dco, dlo, erro, lino, ok = O.ba_step(A, 1e-3, 0)
    assert ok and rel(S, So) < 1e-11 and rel(dc, dco) < 1e-8 and rel(dl, dlo) < 1e-8
    assert np.isclose(err, erro, rtol=1e-12) and not dl[-1].any()


@pytest.mark.parametrize("n_cams", [2, 3, 5, 9, 10, 11, 15, 16, 20, 21, 22, 26, 31, 32, 43])  # n = 6 cams + 5 around the 32- and 64-column edges
def test_block_boundaries_of_the_reduced_system(hip_ctx, n_cams):
    """n = 6 n_cams + 5 against the 32-column diagonal blocks and the 64-column panels: a single partial block (17, 23),
    one real column in the last panel (65), the calibration columns and the right-hand-side row sharing the last panel
    with cameras or getting a panel of their own (n mod 64 > 58: 10, 21, 31 cameras), cameras straddling two panels."""
    sc, A = scene_arrays(seed=20 + n_cams, n_cams=n_cams, n_lm=60 + 12 * n_cams, k=min(n_cams, 5))
    S, g, dc, dl, err, lin = ba.debug_step(hip_ctx, A, 1e-3)
    So, go, dco, dlo, erro, lino, ok = O.ba_step(A, 1e-3, 0)
    assert ok and rel(S, So) < 1e-11 and rel(g, go) < 1e-11
    assert rel(dc, dco) < 1e-8 and rel(dl, dlo) < 1e-8
    assert np.isclose(err, erro, rtol=1e-12) and np.isclose(lin, lino, rtol=1e-9)


@pytest.mark.parametrize("n_cams,n_lm,k", [(12, 200, 6), (31, 400, 6), (60, 600, 8), (130, 2600, 6)])
def test_every_elimination_order_solves_the_same_system(hip_ctx, n_cams, n_lm, k):
    """The reduced camera system is factorised in tiles of the symbolic pattern, level by level of the elimination
    tree of the chosen camera ordering (ba_plan.hpp; the reference gets COLAMD + multifrontal Cholesky through
    SetCeresDefaults, BundleAdjuster.cpp:182-190). The caller's order (one path: the dense chain), reverse
    Cuthill-McKee, nested dissection (independent subtrees in one launch, several sources per target) and the cost
    model's choice must all return the step of the oracle; S itself comes back in the caller's order whatever the layout."""
    sc, A = scene_arrays(seed=30 + n_cams, n_cams=n_cams, n_lm=n_lm, k=k)
    ref = O.ba_step(A, 1e-3, 0)
    assert ref[-1]
    steps = {}
    for name in ("natural", "rcm", "nd", "auto"):
        A.ordering = name
        S, g, dc, dl, err, lin = ba.debug_step(hip_ctx, A, 1e-3)
        assert rel(S, ref[0]) < 1e-11 and rel(g, ref[1]) < 1e-11, name
        assert rel(dc, ref[2]) < 1e-8 and rel(dl, ref[3]) < 1e-8, name
        assert np.isclose(err, ref[4], rtol=1e-12) and np.isclose(lin, ref[5], rtol=1e-9), name
        steps[name] = dc
    for name in ("rcm", "nd", "auto"):
        assert rel(steps[name], steps["natural"]) < 1e-9


def test_unknown_ordering_is_rejected(hip_ctx):
    sc, A = scene_arrays(n_cams=4, n_lm=60, k=3)
    import ctypes as C
    prob, opt = A.c_problem(), ba.c_options(ba.OptimizerConfig.refine_ba())
    prob.ordering = 9
    res = capi.BaResult()
    T = np.zeros((prob.n_cams, 16)); P = np.zeros((prob.n_points, 3))
    res.cam_T_wc, res.points = T.ctypes.data, P.ctypes.data
    assert capi.lib().eacham_ba_solve(hip_ctx.handle, C.byref(prob), C.byref(opt), C.byref(res)) == capi.ERR_INVALID


@pytest.mark.parametrize("ordering", ["natural", "nd"])
def test_lm_run_is_the_same_under_every_ordering(hip_ctx, ordering):
    """A whole LM solve (golden-sized window with outliers): same decisions, same optimum as the oracle."""
    sc, A = scene_arrays(seed=21, n_cams=45, n_lm=1500, k=6)
    A.ordering = ordering
    out = ba.RefineBA(hip_ctx, A, ba.OptimizerConfig.refine_ba())
    ref = O.ba_solve(A, ba.OptimizerConfig.refine_ba())
    assert (out.outer_iterations, out.inner_iterations) == (ref.outer_iterations, ref.inner_iterations)
    assert np.allclose(out.trace, ref.trace, rtol=1e-6)
    assert rel(out.cam_T_wc, ref.cam_T_wc) < 1e-7 and rel(out.points, ref.points) < 1e-7


@pytest.mark.parametrize("lam", [-1e-3, -0.05, -0.5, -1.5])  # the first two: landmark blocks fine, reduced system indefinite
def test_indefinite_system_is_reported_like_the_oracle(hip_ctx, lam):
    """A negative damping makes the damped blocks indefinite at some stage. The diagonal-block factor tests one
    pivot per 32 (a bad pivot poisons every later one with NaN), the oracle tests each: both must agree on
    whether the step exists, and a failed step must surface as an error, never as numbers."""
    sc, A = scene_arrays(seed=8, n_cams=12, n_lm=200, k=6)
    ok = O.ba_step(A, lam, 0)[-1]
    if ok:
        _, _, dc, dl, _, _ = ba.debug_step(hip_ctx, A, lam)
        assert np.isfinite(dc).all() and np.isfinite(dl).all()
    else:
        with pytest.raises(EachamError, match="positive definite"):
            ba.debug_step(hip_ctx, A, lam)


@pytest.mark.parametrize("policy", ["reset", "double"])
@pytest.mark.parametrize("name", ["ba_golden.npz", "ba_golden_hard.npz", "ba_golden_policy.npz"])
def test_golden_fixture(hip_ctx, name, policy):
    """Both readings of LevenbergMarquardtState::decreaseLambda (EACHAM_BA_LM_FACTOR_RESET, the default, and
    _DOUBLE); ba_golden_policy.npz is the fixture whose two traces differ."""
    g = np.load(os.path.join(GOLD_DIR, name))
    pre = "" if policy == "reset" else "double_"
    A = ba.BaArrays(g["cam_T_wc"], g["cam_fixed"], g["points"], g["point_observers"], g["obs_cam"], g["obs_point"],
                    g["obs_uv"], g["K"])
    out = ba.RefineBA(hip_ctx, A, ba.OptimizerConfig("LM", int(g["max_iter"]), float(g["max_toler"]), 10.0, False),
                      lm_factor=policy)
    assert out.outer_iterations == int(g[pre + "outer_iterations"]) and out.inner_iterations == int(g[pre + "inner_iterations"])
    assert np.array_equal(out.trace[:, 3:], g[pre + "trace"][:, 3:])                     # accept/reject decisions
    assert np.allclose(out.trace[:, 0], g[pre + "trace"][:, 0], rtol=1e-6)               # lambda schedule
    acc = g[pre + "trace"][:, 3] == 1
    assert np.allclose(out.trace[acc, 1], g[pre + "trace"][acc, 1], rtol=1e-8)           # nonlinear errors of the accepted steps
    # a rejected try sits at a lambda far too small for its linearisation point: its step is large and its error
    # amplifies the rounding differences of the two solvers (1e-7 seen on the far-off starts); the decision is exact above
    assert np.allclose(out.trace[~acc, 1], g[pre + "trace"][~acc, 1], rtol=1e-5)
    assert rel(out.cam_T_wc, g[pre + "out_T_wc"]) < POSE_POINT_RTOL and rel(out.points, g[pre + "out_points"]) < POSE_POINT_RTOL
    assert rel(out.cam_T_wc, g[pre + "out_T_wc"]) < 1e-7 and rel(out.points, g[pre + "out_points"]) < 1e-7
    # (the far-off starts take 20-50 iterations through ill-conditioned steps: the objective at the end agrees to 1e-8)
    assert np.allclose(out.K, g[pre + "out_K"], rtol=1e-8) and np.isclose(out.final_error, float(g[pre + "final_error"]), rtol=1e-7)


FAULT_LIB = os.path.join(os.path.dirname(capi.LIB_PATH), "exp", "libeacham_hip_fault.so")
FAULT_SCRIPT = """
import sys, numpy as np
from eacham_amd import ba, synth, capi, HipContext, EachamError
A = ba.BaArrays.from_scene(synth.make_scene(60, 400, 8, seed=5))      # n = 365: six panels
with HipContext(0) as ctx:
    try:
        ba.RefineBA(ctx, A, ba.OptimizerConfig(sys.argv[1], 100, 1e-5, 10.0, False))
    except EachamError as e:
        print("code", e.code, e)
        sys.exit(0 if e.code == capi.ERR_HIP and "timed out" in str(e) else 3)
sys.exit(4)
"""


@pytest.mark.parametrize("fault", ["handoff", "progress"])
@pytest.mark.parametrize("method", ["LM", "DogLeg"])
def test_an_expired_in_kernel_wait_is_an_error_not_a_rejected_step(fault, method):
    """The two in-kernel hand-offs of the solve (panel to panel in sp_backsolve, factorising wave to inverting wave
    in the diagonal-block factor) have bounded waits. The diagnostic build
    (-DEXP_BA_FAULT, `make fault`) withholds one signal: the wait must expire, the launch must end, and the solve
    must come back as EACHAM_ERR_HIP — not as 'not positive definite', which LM would answer by silently raising
    lambda. Runs in a child process because the library to load is chosen at import time."""
    import subprocess
    import sys
    assert os.path.exists(FAULT_LIB), "build the diagnostic library with `make -C eacham_amd/csrc fault`"
    env = dict(os.environ, EACHAM_HIP_LIB=FAULT_LIB, EACHAM_FAULT=fault,
               PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", FAULT_SCRIPT, method], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-500:])


def test_dogleg_on_an_unfactorisable_system_reports_indeterminate(hip_ctx):
    """GTSAM's DoglegOptimizer throws IndeterminantLinearSystemException when the Gauss-Newton system cannot be
    factorised; RefineBA would not reach its write-back. Here: status EACHAM_BA_INDETERMINATE and the input values,
    in the oracle and on the device alike. The case: a landmark 1e-160 in front of a camera centre — finite error
    (Huber), Jacobians of 1e163, a landmark block that overflows — with LM on the same window as the contrast
    (every try fails, lambda climbs to its bound, status DONE with zero iterations)."""
    sc, A = scene_arrays(seed=8, n_cams=12, n_lm=200, k=6, outliers=False)
    c, j = int(A.obs_cam[0]), int(A.obs_point[0])
    A.cam_T_wc = A.cam_T_wc.copy()
    A.cam_T_wc[c] = np.eye(4)
    A.points = A.points.copy()
    A.points[j] = [1e-161, 0.0, 1e-160]
    A.obs_uv = A.obs_uv.copy()
    A.obs_uv[0] = [A.K[0] * 0.1 + A.K[2], A.K[3]]
    cfg = ba.OptimizerConfig("DogLeg", 100, 1e-5, 10.0, False)
    out = ba.RefineBA(hip_ctx, A, cfg)
    ref = O.ba_solve(A, cfg)
    assert out.status == ref.status == capi.BA_INDETERMINATE
    assert out.outer_iterations == ref.outer_iterations == 0 and np.isfinite(out.initial_error)
    assert np.isclose(out.initial_error, ref.initial_error, rtol=1e-12)
    assert np.array_equal(out.points, A.points) and rel(out.cam_T_wc.reshape(-1, 16), A.cam_T_wc.reshape(-1, 16)) < 1e-15
    lm, lm_ref = ba.RefineBA(hip_ctx, A, ba.OptimizerConfig.refine_ba()), O.ba_solve(A, ba.OptimizerConfig.refine_ba())
    assert lm.status == lm_ref.status == capi.BA_DONE and lm.outer_iterations == lm_ref.outer_iterations == 0
    assert lm.inner_iterations == lm_ref.inner_iterations and not lm.trace[:, 3].any()


def test_unknown_factor_policy_is_rejected(hip_ctx):
    sc, A = scene_arrays(n_cams=4, n_lm=60, k=3)
    L = capi.lib()
    import ctypes as C
    prob, opt = A.c_problem(), ba.c_options(ba.OptimizerConfig.refine_ba())
    opt.lm_factor_policy = 7
    res = capi.BaResult()
    T = np.zeros((prob.n_cams, 16)); P = np.zeros((prob.n_points, 3))
    res.cam_T_wc, res.points = T.ctypes.data, P.ctypes.data
    assert L.eacham_ba_solve(hip_ctx.handle, C.byref(prob), C.byref(opt), C.byref(res)) == capi.ERR_INVALID


def test_pcg_block_jacobi_parity(hip_ctx):
    """usePreconditioner (BundleAdjuster.cpp:192-200): every damped system solved by PCG + block-Jacobi at 1e-10, on the
    device in block form, in the oracle in Jacobian form (different roundings of the same operator, so the CG iterates
    are not bit-identical). Same LM decisions, poses / points within the north-star tolerance of the oracle's PCG run
    AND of the direct solve; the iteration counts of the inner solver agree to a few per cent."""
    for A in (scene_arrays(seed=21, n_cams=12, n_lm=900, k=5)[1],
              ba.BaArrays.from_scene(synth.local_window(synth.make_scene(120, 7200, 10, seed=3), 60))):
        cfg = ba.OptimizerConfig("LM", 100, 1e-5, 10.0, True)
        out = ba.RefineBA(hip_ctx, A, cfg)
        ref = O.ba_solve(A, cfg)
        direct = ba.RefineBA(hip_ctx, A, ba.OptimizerConfig("LM", 100, 1e-5, 10.0, False))
        assert out.status == ref.status == 0 and out.reserved > 20 * out.inner_iterations and direct.reserved == 0
        assert abs(out.reserved - ref.reserved) <= 0.05 * ref.reserved + 5
        assert (out.outer_iterations, out.inner_iterations) == (ref.outer_iterations, ref.inner_iterations)
        assert np.array_equal(out.trace[:, 3:], ref.trace[:, 3:]) and np.allclose(out.trace[:, :3], ref.trace[:, :3], rtol=1e-5)
        for other in (ref, direct):
            assert rel(out.cam_T_wc, other.cam_T_wc) < POSE_POINT_RTOL and rel(out.points, other.points) < POSE_POINT_RTOL
            assert np.isclose(out.final_error, other.final_error, rtol=1e-7)
    # DogLeg ignores the flag (the reference sets it inside its LM branch only)
    sc, A = scene_arrays(seed=21, n_cams=12, n_lm=900, k=5)
    a = ba.RefineBA(hip_ctx, A, ba.OptimizerConfig("DogLeg", 100, 1e-5, 10.0, True))
    b = ba.RefineBA(hip_ctx, A, ba.OptimizerConfig("DogLeg", 100, 1e-5, 10.0, False))
    assert a.reserved == 0 and np.array_equal(a.points, b.points)


@pytest.mark.parametrize("cfg", [ba.OptimizerConfig.refine_ba(), ba.OptimizerConfig.global_ba()])
def test_refine_ba_parity(hip_ctx, cfg):
    sc, A = scene_arrays(seed=21, n_cams=12, n_lm=900, k=5)
    out = ba.RefineBA(hip_ctx, A, cfg)
    ref = O.ba_solve(A, cfg)
    assert out.status == ref.status == 0
    assert (out.outer_iterations, out.inner_iterations) == (ref.outer_iterations, ref.inner_iterations)
    assert out.outer_iterations <= cfg.maxIter
    assert np.allclose(out.trace, ref.trace, rtol=1e-6)
    assert rel(out.cam_T_wc, ref.cam_T_wc) < POSE_POINT_RTOL and rel(out.points, ref.points) < POSE_POINT_RTOL
    assert rel(out.cam_T_wc, ref.cam_T_wc) < 1e-7 and rel(out.points, ref.points) < 1e-7
    assert np.allclose(out.K, ref.K, rtol=1e-9)
    assert np.isclose(out.initial_error, ref.initial_error, rtol=1e-12)
    assert np.isclose(out.final_error, ref.final_error, rtol=1e-9)
    # camera 0 is the fixed node: it must not move (prior sigma 1e-4)
    assert np.abs(out.cam_T_wc[0] - A.cam_T_wc.reshape(-1, 4, 4)[0]).max() < 1e-5


def test_hard_start_with_rejected_steps(hip_ctx):
    sc = synth.make_scene(6, 90, 2, seed=1, rot_noise=0.5, trans_noise=0.5, point_noise=0.8)
    A = ba.BaArrays.from_scene(sc)
    out = ba.RefineBA(hip_ctx, A, ba.OptimizerConfig.refine_ba())
    ref = O.ba_solve(A, ba.OptimizerConfig.refine_ba())
    assert (ref.trace[:, 3] == 0).sum() >= 3  # the case does exercise increaseLambda, several times
    # 21 iterations through rejected steps from a far-off start: the decisions are identical, the numbers carry the
    # amplified rounding differences of the two solvers (lambda follows the step quality: 5e-6; rejected tries: 1e-6)
    acc = ref.trace[:, 3] == 1
    assert np.array_equal(out.trace[:, 3:], ref.trace[:, 3:]) and np.allclose(out.trace[:, 0], ref.trace[:, 0], rtol=1e-4)
    assert np.allclose(out.trace[acc, 1], ref.trace[acc, 1], rtol=1e-7) and np.allclose(out.trace[~acc, 1], ref.trace[~acc, 1], rtol=1e-4)
    assert rel(out.points, ref.points) < POSE_POINT_RTOL and rel(out.cam_T_wc, ref.cam_T_wc) < POSE_POINT_RTOL


@pytest.mark.parametrize("delta", [10.0, 0.05])
def test_dogleg_parity(hip_ctx, delta):
    """method "DogLeg" (BundleAdjuster.cpp:204-214): GTSAM's DoglegOptimizer control flow; the trace
    carries (trust radius, f(x_d), model decrease, accepted, outer)."""
    sc, A = scene_arrays(seed=21, n_cams=12, n_lm=900, k=5)
    cfg = ba.OptimizerConfig("DogLeg", 100, 1e-5, delta, False)
    out = ba.RefineBA(hip_ctx, A, cfg)
    ref = O.ba_solve(A, cfg)
    lm = O.ba_solve(A, ba.OptimizerConfig.refine_ba())
    assert out.status == ref.status == 0
    assert (out.outer_iterations, out.inner_iterations) == (ref.outer_iterations, ref.inner_iterations)
    assert np.array_equal(out.trace[:, 3:], ref.trace[:, 3:]) and np.allclose(out.trace[:, :3], ref.trace[:, :3], rtol=1e-6)
    assert rel(out.cam_T_wc, ref.cam_T_wc) < 1e-7 and rel(out.points, ref.points) < 1e-7
    assert np.allclose(out.K, ref.K, rtol=1e-9) and np.isclose(out.final_error, ref.final_error, rtol=1e-9)
    assert np.isclose(out.final_lambda, ref.final_lambda, rtol=1e-6)          # the final trust radius
    assert np.isclose(out.final_error, lm.final_error, rtol=1e-4)             # same optimum as LM
    if delta < 1:
        assert out.trace[0, 0] == np.float32(delta) and out.trace[1, 0] > out.trace[0, 0]  # the region grows by 3 |x_d|


def test_dogleg_hard_start_shrinks_the_region(hip_ctx):
    sc = synth.make_scene(6, 90, 2, seed=0, rot_noise=0.5, trans_noise=0.5, point_noise=0.8)
    A = ba.BaArrays.from_scene(sc)
    cfg = ba.OptimizerConfig("DogLeg", 100, 1e-5, 10.0, False)
    out = ba.RefineBA(hip_ctx, A, cfg)
    ref = O.ba_solve(A, cfg)
    assert (ref.trace[:, 3] == 0).sum() >= 1  # the case does exercise the rho < 0 branch
    assert np.array_equal(out.trace[:, 3:], ref.trace[:, 3:]) and np.allclose(out.trace[:, :2], ref.trace[:, :2], rtol=1e-6)
    assert rel(out.points, ref.points) < POSE_POINT_RTOL and rel(out.cam_T_wc, ref.cam_T_wc) < POSE_POINT_RTOL


def test_problems_share_the_arena_pool_without_seeing_each_other(hip_ctx):
    """Every prepared problem lives in one arena taken from the context's pool and gives it back on release: problems
    of different sizes alive at the same time, released out of order and followed by new ones that reuse their arenas
    (dirty memory) must solve exactly as they do alone."""
    cfg = ba.OptimizerConfig.refine_ba()
    scenes = [scene_arrays(seed=21 + k, n_cams=nc, n_lm=nl, k=6)[1] for k, (nc, nl) in enumerate([(6, 120), (25, 500), (12, 260), (40, 300)])]
    alone = []
    for A in scenes:
        P = ba.PreparedBA(hip_ctx, A)
        alone.append(P.run(cfg, trace_cap=0))
        P.close()
    live = [ba.PreparedBA(hip_ctx, A) for A in scenes[:3]]      # three arenas busy at once
    outs = [None] * 4
    outs[1] = live[1].run(cfg, trace_cap=0)
    outs[0] = live[0].run(cfg, trace_cap=0)
    live[1].close()                                             # the big one goes back first ...
    late = ba.PreparedBA(hip_ctx, scenes[3])                    # ... and is taken by a problem of another shape
    outs[2] = live[2].run(cfg, trace_cap=0)
    outs[3] = late.run(cfg, trace_cap=0)
    again = live[0].run(cfg, trace_cap=0)                       # a handle solves from its uploaded start every time
    for P in (live[0], live[2], late):
        P.close()
    for got, ref in zip(outs + [again], alone + [alone[0]]):
        assert got.outer_iterations == ref.outer_iterations and got.inner_iterations == ref.inner_iterations
        assert got.final_error == ref.final_error
        assert np.array_equal(got.cam_T_wc, ref.cam_T_wc) and np.array_equal(got.points, ref.points)


def test_fewer_than_50_landmarks_is_a_silent_no_op(hip_ctx):
    sc, A = scene_arrays(n_cams=4, n_lm=49, k=3, outliers=False)
    out = ba.RefineBA(hip_ctx, A, ba.OptimizerConfig.refine_ba())
    assert out.status == capi.BA_SKIPPED and out.outer_iterations == 0
    assert np.allclose(out.points, A.points, atol=0) and rel(out.cam_T_wc.reshape(-1, 16), A.cam_T_wc.reshape(-1, 16)) < 1e-15


def test_errors(hip_ctx):
    sc, A = scene_arrays(n_cams=4, n_lm=60, k=3)
    A.obs_cam = A.obs_cam.copy()
    A.obs_cam[0] = 99
    with pytest.raises(EachamError) as e:
        ba.RefineBA(hip_ctx, A, ba.OptimizerConfig.refine_ba())
    assert e.value.code == capi.ERR_INVALID


def test_metric_scene_s200(hip_ctx):
    """BASELINE metric scene: 200 cameras / 50k landmarks / 500k observations, refine_ba options."""
    sc = synth.make_scene(200, 50_000, 10)
    A = ba.BaArrays.from_scene(sc)
    solver = ba.PreparedBA(hip_ctx, A)
    out = solver.run(ba.OptimizerConfig.refine_ba())
    again = solver.run(ba.OptimizerConfig.refine_ba())     # deterministic: no atomics anywhere
    solver.close()
    assert np.array_equal(out.points, again.points) and np.array_equal(out.trace, again.trace)
    ref = O.ba_solve(A, ba.OptimizerConfig.refine_ba())
    assert (out.outer_iterations, out.inner_iterations) == (ref.outer_iterations, ref.inner_iterations)
    assert np.allclose(out.trace, ref.trace, rtol=1e-6)
    assert rel(out.cam_T_wc, ref.cam_T_wc) < POSE_POINT_RTOL and rel(out.points, ref.points) < POSE_POINT_RTOL
    # size-independent properties: the error drops by orders of magnitude and the truth is recovered
    assert out.final_error < 0.05 * out.initial_error
    assert np.abs(out.points - sc["points_true"]).max() < 0.05
    assert np.abs(out.K - sc["K"]).max() < 5.0


def test_config4_size_properties(hip_ctx):
    """BASELINE configs[3]: 500 cams / 100k landmarks / 1M observations (n = 3005: 47 panels in the caller's order,
    ~59 under nested dissection with an elimination tree 16 levels high). Too large for the oracle to finish in seconds, so
    the check is by size-independent properties: determinism, a large error drop, recovery of the truth, and
    agreement of LM and DogLeg on the optimum."""
    sc = synth.make_scene(500, 100_000, 10, seed=4)
    A = ba.BaArrays.from_scene(sc)
    solver = ba.PreparedBA(hip_ctx, A)
    out = solver.run(ba.OptimizerConfig.refine_ba())
    again = solver.run(ba.OptimizerConfig.refine_ba())
    dl = solver.run(ba.OptimizerConfig("DogLeg", 100, 1e-5, 10.0, False))
    solver.close()
    assert out.status == 0 and 2 <= out.outer_iterations <= 20
    assert np.array_equal(out.points, again.points) and np.array_equal(out.cam_T_wc, again.cam_T_wc)
    assert out.final_error < 0.05 * out.initial_error
    err = np.linalg.norm(out.points - sc["points_true"], axis=1)
    err0 = np.linalg.norm(sc["points_init"] - sc["points_true"], axis=1)
    assert np.median(err) < 0.35 * np.median(err0), (np.median(err), np.median(err0), err.max())
    assert np.isclose(dl.final_error, out.final_error, rtol=1e-4)
    assert (np.diff(out.trace[out.trace[:, 3] == 1, 1]) < 0).all()   # accepted steps only ever lower the error


def _ctx_with(**env):
    """A context of its own created under the given environment switches (they are read once, at eacham_ctx_create)."""
    from eacham_amd import HipContext
    old = {k: os.environ.get(k) for k in env}
    try:
        os.environ.update(env)
        return HipContext(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("prepare", ["host", "device"])
def test_both_forms_of_the_schur_stage_solve_the_same_system(hip_ctx, prepare):
    """The landmark-major form (ba_schur_groups: Et in LDS, one partial per segment of a camera block) and the pair-list form of
    rounds 1-4 (Et through HBM) add the same products in different orders: the reduced system, the step and a whole LM run agree
    to rounding, both with the oracle; group sizes other than the default change the order again, not the answer."""
    sc, A = scene_arrays(seed=31, n_cams=40, n_lm=1800, k=7, pixel_noise=1.0)
    cfg = ba.OptimizerConfig.refine_ba()
    ref = O.ba_solve(A, cfg)
    So, go, dco, dlo, erro, lino, ok = O.ba_step(A, 1e-3, 0)
    outs = []
    for env in [dict(EACHAM_BA_SCHUR="groups"), dict(EACHAM_BA_SCHUR="pairs"), dict(EACHAM_BA_SCHUR="groups", EACHAM_BA_GROUP_ROWS="64"),
                dict(EACHAM_BA_SCHUR="groups", EACHAM_BA_GROUP_ROWS="256"), dict(EACHAM_BA_SCHUR="groups", EACHAM_BA_GROUP_ROWS="480")]:
        ctx = _ctx_with(EACHAM_BA_PREPARE=prepare, **env)
        try:
            pb = ba.PreparedBA(ctx, A)
            groups = pb.structure("g_groups").reshape(-1, 8)
            assert (len(groups) == 0) == (env.get("EACHAM_BA_SCHUR") == "pairs") and (len(pb.structure("blocks")) == 0) == (len(groups) > 0)
            if "EACHAM_BA_GROUP_ROWS" in env:
                assert groups[:, 3].max() <= int(env["EACHAM_BA_GROUP_ROWS"]) and groups[:, 1].max() <= int(env["EACHAM_BA_GROUP_ROWS"]) // 4
            pb.close()
            S, g, dc, dl, err, lin = ba.debug_step(ctx, A, 1e-3)
            assert rel(S, So) < 1e-11 and rel(g, go) < 1e-11 and rel(dc, dco) < 1e-8 and rel(dl, dlo) < 1e-8
            out = ba.RefineBA(ctx, A, cfg)
            assert (out.outer_iterations, out.inner_iterations) == (ref.outer_iterations, ref.inner_iterations)
            assert np.array_equal(out.trace[:, 3:], ref.trace[:, 3:]) and np.allclose(out.trace[:, :2], ref.trace[:, :2], rtol=1e-6)
            assert rel(out.cam_T_wc, ref.cam_T_wc) < 1e-7 and rel(out.points, ref.points) < 1e-7
            again = ba.RefineBA(ctx, A, cfg)     # and bit for bit from run to run
            assert np.array_equal(out.points, again.points) and np.array_equal(out.cam_T_wc, again.cam_T_wc) and np.array_equal(out.trace, again.trace)
            outs.append(out)
        finally:
            ctx.close()
    for o in outs[1:]:
        assert rel(o.points, outs[0].points) < 1e-9 and rel(o.cam_T_wc, outs[0].cam_T_wc) < 1e-9


def test_the_dense_form_of_a_local_window_solves_the_same_system(hip_ctx):
    """The dense form of the Schur stage for a local window (csrc/ba_window.hpp, EACHAM_BA_SCHUR=dense: a workgroup forms its share
    of EVERY block of the reduced system straight from the values, no linearisation launch and no pair lists; measured slower than
    the pair lists on the TUM stand-in's windows, hence not the default): the reduced system, the step and whole LM runs agree
    with the oracle and with the pair-list form to rounding, whatever the group size; the structure is the one ba_window.hpp
    defines; a problem the form does not cover (a camera seeing a landmark twice, more than 24 cameras, DogLeg) takes the pair lists."""
    scene = synth.make_scene(60, 4000, 8, seed=5, pixel_noise=1.0)
    W = synth.local_window(scene, 30, max_neighbours=18)
    A = ba.BaArrays.from_scene(W)
    A.obs_uv[::17] += 25.0
    assert A.cam_T_wc.shape[0] == 19
    cfg = ba.OptimizerConfig.refine_ba()
    ref = O.ba_solve(A, cfg)
    So, go, dco, dlo, erro, lino, ok = O.ba_step(A, 1e-3, 0)
    outs = []
    for env in [dict(EACHAM_BA_SCHUR="pairs"), dict(EACHAM_BA_SCHUR="dense"), dict(EACHAM_BA_SCHUR="dense", EACHAM_BA_WINDOW_ROWS="96"),   # (64-row groups: 12.5 MB of partials, beyond the form's 12 MB bound)
                dict(EACHAM_BA_SCHUR="dense", EACHAM_BA_WINDOW_ROWS="200"), dict()]:   # (256 rows of a 19-camera window need 167 KB of LDS: not built)
        ctx = _ctx_with(**env)
        try:
            if env.get("EACHAM_BA_SCHUR") == "dense":
                pb = ba.PreparedBA(ctx, A)
                groups, lmid = pb.structure("w_groups").reshape(-1, 2), pb.structure("w_lmid")
                rows = len(pb.structure("w_rowinfo")) // (2 * len(groups))
                assert rows == int(env.get("EACHAM_BA_WINDOW_ROWS", rows)) and groups[:, 1].max() <= rows and groups[:, 0].max() <= rows // 4
                ri = pb.structure("w_rowinfo").reshape(len(groups), rows, 2)
                used = np.unique(A.obs_point)
                assert np.array_equal(lmid[lmid >= 0], used) and len(pb.structure("blocks")) == 0   # every landmark once, in order; no pair lists
                assert groups[:, 1].sum() == len(A.obs_cam) + len(used) and (ri[:, :, 0] == 19).sum() == len(used)
                for g in range(len(groups)):   # a landmark's rows: ascending cameras, then its own row
                    cams = ri[g, :groups[g, 1], 0]
                    assert ((np.diff(cams) > 0) | (cams[:-1] == 19)).all() and cams[-1] == 19 and (ri[g, groups[g, 1]:, 0] == -1).all()
                with pytest.raises(EachamError):   # the form carries the direct LM solve only
                    pb.run(ba.OptimizerConfig(method="DogLeg", maxIter=5, maxTolerance=1e-5, delta=1.0, usePreconditioner=False))
                pb.close()
            if env:
                S, g, dc, dl, err, lin = ba.debug_step(ctx, A, 1e-3)
                assert rel(S, So) < 1e-11 and rel(g, go) < 1e-11 and rel(dc, dco) < 1e-8 and rel(dl, dlo) < 1e-8
                assert np.isclose(err, erro, rtol=1e-12) and np.isclose(lin, lino, rtol=1e-9)
            out = ba.RefineBA(ctx, A, cfg)
            assert (out.outer_iterations, out.inner_iterations) == (ref.outer_iterations, ref.inner_iterations)
            assert np.array_equal(out.trace[:, 3:], ref.trace[:, 3:]) and np.allclose(out.trace[:, :2], ref.trace[:, :2], rtol=1e-6)
            assert rel(out.cam_T_wc, ref.cam_T_wc) < 1e-7 and rel(out.points, ref.points) < 1e-7 and rel(out.K, ref.K) < 1e-9
            again = ba.RefineBA(ctx, A, cfg)     # and bit for bit from run to run
            assert np.array_equal(out.points, again.points) and np.array_equal(out.cam_T_wc, again.cam_T_wc) and np.array_equal(out.trace, again.trace)
            outs.append(out)
        finally:
            ctx.close()
    for o in outs[1:]:
        assert rel(o.points, outs[0].points) < 1e-9 and rel(o.cam_T_wc, outs[0].cam_T_wc) < 1e-9
    assert np.array_equal(outs[0].points, outs[4].points)   # the default is the pair lists
    # not covered: a camera that sees a landmark twice, 27 cameras — the pair lists serve (DogLeg on a dense handle is refused, above)
    B = ba.BaArrays.from_scene(W)
    B.obs_cam = np.append(B.obs_cam, B.obs_cam[0]).astype(np.uint32)
    B.obs_point = np.append(B.obs_point, B.obs_point[0]).astype(np.uint32)
    B.obs_uv = np.vstack([B.obs_uv, B.obs_uv[:1] + 0.5])
    bref = O.ba_solve(B, cfg)
    ctx = _ctx_with(EACHAM_BA_SCHUR="dense")
    try:
        for arrays in (B, ba.BaArrays.from_scene(synth.local_window(scene, 30, min_shared=3, max_neighbours=26))):
            pb = ba.PreparedBA(ctx, arrays)
            assert len(pb.structure("w_groups")) == 0 and len(pb.structure("blocks")) > 0
            pb.close()
        bout = ba.RefineBA(ctx, B, cfg)
        assert bout.inner_iterations == bref.inner_iterations and rel(bout.points, bref.points) < 1e-7
    finally:
        ctx.close()


@pytest.mark.parametrize("prepare", ["host", "device"])
def test_a_landmark_too_heavy_for_a_group_takes_the_pair_lists(prepare):
    """A landmark seen by 90 of 100 cameras has 4186 entries: more than a 128-row group may hold (2048). The problem then keeps
    the pair lists of rounds 1-4 (ba_groups.hpp step 2) — same answer as the oracle, no group structure."""
    sc = synth.make_scene(100, 400, 6, seed=41, pixel_noise=1.0)
    A = ba.BaArrays.from_scene(sc)
    cams = np.arange(90, dtype=np.uint32)
    T = A.cam_T_wc[cams]
    X = np.append(A.points[7], 1.0)
    pc = np.einsum("nij,j->ni", T, X)
    uv = np.stack([A.K[0] * pc[:, 0] / pc[:, 2] + A.K[2], A.K[1] * pc[:, 1] / pc[:, 2] + A.K[3]], 1)
    keep = (A.obs_point != 7) & np.ones(len(A.obs_point), bool)
    A.obs_cam = np.concatenate([A.obs_cam[keep], cams]).astype(np.uint32)
    A.obs_point = np.concatenate([A.obs_point[keep], np.full(90, 7)]).astype(np.uint32)
    A.obs_uv = np.concatenate([A.obs_uv[keep], uv.astype(np.float32).astype(np.float64)])
    A.point_observers[7] = 90
    cfg = ba.OptimizerConfig.refine_ba()
    ref = O.ba_solve(A, cfg)
    ctx = _ctx_with(EACHAM_BA_PREPARE=prepare, EACHAM_BA_SCHUR="groups")
    try:
        pb = ba.PreparedBA(ctx, A)
        assert len(pb.structure("g_groups")) == 0 and len(pb.structure("blocks")) > 0
        pb.close()
        out = ba.RefineBA(ctx, A, cfg)
        assert (out.outer_iterations, out.inner_iterations) == (ref.outer_iterations, ref.inner_iterations)
        assert rel(out.cam_T_wc, ref.cam_T_wc) < 1e-7 and rel(out.points, ref.points) < 1e-7
    finally:
        ctx.close()
