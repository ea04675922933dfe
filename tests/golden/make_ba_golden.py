"""Generates tests/golden/ba_golden.npz with the CPU oracle (oracle/ba_oracle.c).

The reference has no golden vectors for this path and GTSAM cannot be built here (SURVEY.md §8c:
parity unpinned), so the fixture records the repo's own restatement: inputs, the LM trace
(lambda / error / linear cost change / accepted per tryLambda call) and the final values.
Options chosen: first-order Pose3 chart with Cayley Rot3, Ceres-default LM.
Run from the repo root:  python tests/golden/make_ba_golden.py
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from eacham_amd import ba, synth  # noqa: E402
import oracle_api as O  # noqa: E402


def main():
    make("ba_golden.npz", synth.make_scene(6, 90, 4, seed=synth.MASTER_SEED, pixel_noise=1.0))
    # far-off start: the trace contains rejected steps (lambda increases) as well
    make("ba_golden_hard.npz", synth.make_scene(6, 90, 2, seed=0, rot_noise=0.5, trans_noise=0.5, point_noise=0.8))
    # a far-off start whose trace has reject -> accept -> reject sequences: the two readings of
    # LevenbergMarquardtState::decreaseLambda (EACHAM_BA_LM_FACTOR_RESET / _DOUBLE) give different lambda
    # schedules, iterates and iteration counts here; both are recorded
    make("ba_golden_policy.npz", synth.make_scene(6, 90, 2, seed=1, rot_noise=0.5, trans_noise=0.5, point_noise=0.8),
         perturb=False)


def make(name, sc, perturb=True):
    A = ba.BaArrays.from_scene(sc)
    if perturb:
        A.obs_uv[::13] += 30.0  # a few outliers so the Huber branch is in the fixture
        A.point_observers[::5] += 2  # global observer counts exceed the window's (local BA)
    cfg = ba.OptimizerConfig.refine_ba()
    out = O.ba_solve(A, cfg, nthreads=1, lm_factor="reset")    # the default policy: unprefixed keys
    alt = O.ba_solve(A, cfg, nthreads=1, lm_factor="double")
    rev = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", name),
        cam_T_wc=A.cam_T_wc, cam_fixed=A.cam_fixed, points=A.points, point_observers=A.point_observers,
        obs_cam=A.obs_cam, obs_point=A.obs_point, obs_uv=A.obs_uv, K=A.K,
        max_iter=np.int32(cfg.maxIter), max_toler=np.float32(cfg.maxTolerance),
        trace=out.trace, out_T_wc=out.cam_T_wc, out_points=out.points, out_K=out.K,
        initial_error=out.initial_error, final_error=out.final_error,
        outer_iterations=out.outer_iterations, inner_iterations=out.inner_iterations,
        double_trace=alt.trace, double_out_T_wc=alt.cam_T_wc, double_out_points=alt.points, double_out_K=alt.K,
        double_final_error=alt.final_error, double_outer_iterations=alt.outer_iterations,
        double_inner_iterations=alt.inner_iterations,
        generator=np.array(f"oracle/ba_oracle.c @ {rev}; LM Ceres defaults, Cayley/first-order Pose3 chart, refine_ba (100, 1e-5); "
                           "unprefixed keys: lm_factor_policy RESET (2 * lambdaFactor), double_*: DOUBLE (2 * currentFactor)"))
    print(name, out.initial_error, out.final_error, out.outer_iterations, out.inner_iterations,
          "rejected:", int((out.trace[:, 3] == 0).sum()), "| double:", alt.final_error, alt.outer_iterations, alt.inner_iterations,
          "same trace:", out.trace.shape == alt.trace.shape and bool(np.allclose(out.trace, alt.trace)))


if __name__ == "__main__":
    main()
