"""CPU: the minimal solvers of the robust estimators (oracle/solve_oracle.c, test infrastructure) against independent numpy
statements and against ground truth.
  homography4  OpenCV 4.5.5 HomographyEstimatorCallback::runKernel  <- cv::findHomography, ReconstructionManager.cpp:75
  essential5   EMEstimatorCallback::runKernel (Nister five-point)   <- cv::findEssentialMat, ReconstructionManager.cpp:57-61
PARITY UNPINNED (OpenCV is not in the image and its sampling is tied to its RNG): the checks are mathematical —
the plain SVD form of the DLT, the defining equations of an essential matrix, the true model among the solutions."""
import numpy as np
import pytest

import oracle_api as O
import score_cases as SC


def dlt_numpy(a, b):
    """Unnormalised 4-point DLT by SVD: the same H up to scale (an independent statement of the same null vector)."""
    rows = []
    for (X, Y), (x, y) in zip(a, b):
        rows.append([X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x])
        rows.append([0, 0, 0, X, Y, 1, -y * X, -y * Y, -y])
    h = np.linalg.svd(np.array(rows))[2][-1]
    return (h / h[8]).reshape(3, 3)


def test_homography4_is_the_dlt_null_vector_and_recovers_the_plane_homography():
    c = SC.two_view_case(n=400, seed=3, outliers=0.0, planar=True)
    rng = np.random.default_rng(5)
    samples = np.array([rng.choice(400, 4, replace=False) for _ in range(200)], np.int32)
    models, counts = O.solve_minimal("homography4", c["uv1"], c["uv2"], samples)
    assert counts.min() == 1 and np.allclose(models[:, 0, 8], 1.0)
    worst = 0.0
    for s, H in zip(samples, models[:, 0].reshape(-1, 3, 3)):
        a, b = c["uv1"][s], c["uv2"][s]
        p = np.c_[a, np.ones(4)] @ H.T
        assert np.abs(p[:, :2] / p[:, 2:] - b).max() < 1e-6          # the four correspondences are mapped exactly
        Hn = dlt_numpy(a, b)
        worst = max(worst, np.abs(H - Hn).max() / np.abs(Hn).max())
    assert worst < 1e-6                                                  # (the unnormalised DLT is the worse conditioned of the two)
    # noise-free pixels of a plane: every sample gives THE homography
    c = SC.two_view_case(n=60, seed=4, outliers=0.0, planar=True)
    sc_h = c["H"][0].reshape(3, 3)
    X = np.c_[c["uv1"], np.ones(60)] @ sc_h.T
    exact2 = X[:, :2] / X[:, 2:]
    samples = np.array([rng.choice(60, 4, replace=False) for _ in range(50)], np.int32)
    models, counts = O.solve_minimal("homography4", c["uv1"], exact2, samples)
    assert counts.min() == 1 and np.abs(models[:, 0].reshape(-1, 3, 3) - sc_h).max() / np.abs(sc_h).max() < 1e-7


def test_homography4_degenerate_samples_are_reported():
    a = np.array([[1.0, 1], [2, 1], [3, 1], [4, 1]])                     # no spread in y: OpenCV's runKernel returns 0 models
    models, counts = O.solve_minimal("homography4", a, a + 1.0, np.array([[0, 1, 2, 3]], np.int32))
    assert counts[0] == 0 and not models.any()


def essential_residuals(E, x1, x2):
    x1h, x2h = np.c_[x1, np.ones(len(x1))], np.c_[x2, np.ones(len(x2))]
    epi = np.abs(np.einsum("ni,ij,nj->n", x2h, E, x1h)).max()
    cubic = np.abs(2 * E @ E.T @ E - np.trace(E @ E.T) * E).max()
    return epi, cubic, abs(np.linalg.det(E))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_essential5_solutions_are_essential_matrices_and_contain_the_truth(seed):
    c = SC.two_view_case(n=300, seed=seed, outliers=0.0)
    K = c["K"]
    rng = np.random.default_rng(seed)
    # exact correspondences of the true relative pose: re-project through E's epipolar geometry is not needed — take the
    # noise-free pixels of the scene by recomputing them from the truth stored in the case
    sc = SC.synth.make_scene(2, 300, 2, seed=seed, pixel_noise=0.0)
    T1, T2, X = sc["T_true"][0], sc["T_true"][1], sc["points_true"]

    def proj(T):
        pc = X @ T[:3, :3].T + T[:3, 3]
        return np.stack([K[0] * pc[:, 0] / pc[:, 2] + K[2], K[1] * pc[:, 1] / pc[:, 2] + K[3]], 1)
    uv1, uv2 = proj(T1), proj(T2)
    T21 = T2 @ np.linalg.inv(T1)
    Et = SC.skew(T21[:3, 3]) @ T21[:3, :3]
    Et /= np.linalg.norm(Et)
    samples = np.array([rng.choice(300, 5, replace=False) for _ in range(300)], np.int32)
    models, counts = O.solve_minimal("essential5", uv1, uv2, samples, K)
    assert counts.min() >= 1 and counts.max() <= 10 and (counts % 2 == 0).all()   # real roots of a real degree-10 polynomial come in pairs
    n1 = (uv1 - K[2:]) / K[:2]
    n2 = (uv2 - K[2:]) / K[:2]
    found = 0
    for s, Es, n in zip(samples, models.reshape(-1, 10, 3, 3), counts):
        assert not Es[n:].any()
        for E in Es[:n]:
            epi, cubic, det = essential_residuals(E, n1[s], n2[s])
            assert epi < 1e-9 and cubic < 1e-8 and det < 1e-9 and abs(np.linalg.norm(E) - 1) < 1e-12
        found += min(min(np.abs(E - Et).max(), np.abs(E + Et).max()) for E in Es[:n]) < 1e-7
    assert found == len(samples)                                        # the true E is one of the solutions of every sample
    # with pixels instead of normalised coordinates and K = None nothing matches (the caller must say which it passes)
    m2, c2 = O.solve_minimal("essential5", n1, n2, samples[:20], None)
    assert np.array_equal(c2, counts[:20]) and np.allclose(m2, models[:20], atol=1e-9)


def test_essential5_noisy_samples_and_degenerate_input():
    c = SC.two_view_case(n=500, seed=9, outliers=0.2)
    rng = np.random.default_rng(9)
    samples = np.array([rng.choice(500, 5, replace=False) for _ in range(400)], np.int32)
    models, counts = O.solve_minimal("essential5", c["uv1"], c["uv2"], samples, c["K"])
    assert (counts % 2 == 0).all() and counts.max() <= 10 and counts.mean() > 2
    n1 = (c["uv1"] - c["K"][2:]) / c["K"][:2]
    n2 = (c["uv2"] - c["K"][2:]) / c["K"][:2]
    for s, Es, n in zip(samples, models.reshape(-1, 10, 3, 3), counts):
        for E in Es[:n]:
            epi, cubic, det = essential_residuals(E, n1[s], n2[s])
            assert epi < 1e-7 and cubic < 1e-6
    same = np.zeros((5, 2))                                              # five identical points: no model, no crash
    m, k = O.solve_minimal("essential5", same, same, np.array([[0, 1, 2, 3, 4]], np.int32), None)
    assert k[0] == 0 and not m.any()


def exact_pnp(n=300, seed=11):
    """Object points, their EXACT pixels under a known pose (no noise, no rounding to float32), K and that pose."""
    c = SC.pnp_case(n=n, seed=seed, outliers=0.0)
    X, T, K = np.ascontiguousarray(c["X"][1:]), c["models"][0], np.asarray(c["K"], float)
    pc = X @ T[:9].reshape(3, 3).T + T[9:]
    uv = np.stack([K[0] * pc[:, 0] / pc[:, 2] + K[2], K[1] * pc[:, 1] / pc[:, 2] + K[3]], 1)
    return X, uv, K, T


def test_epnp_recovers_the_pose_exactly_and_averages_noise_down():
    """EPnP (cv::solvePnPRansac(..., SOLVEPNP_EPNP), ReconstructionManager.cpp:227-228): exact pixels -> the exact pose from
    5 points up to all of them; noisy pixels -> an error that shrinks with the sample size (the all-inlier refit)."""
    X, uv, K, T = exact_pnp()
    rng = np.random.default_rng(2)
    for m in (5, 6, 12, 299):
        samples = np.array([rng.choice(len(X), m, replace=False) for _ in range(100)], np.int32)
        models, ok = O.solve_pnp(X, uv, K, samples)
        assert ok.all() and np.abs(models - T).max() < 1e-9
        R = models[:, :9].reshape(-1, 3, 3)
        assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-12 and np.abs(np.linalg.det(R) - 1).max() < 1e-12
    noisy = uv + 0.5 * rng.normal(size=uv.shape)
    med = []
    for m in (5, 20, 299):
        samples = np.array([rng.choice(len(X), m, replace=False) for _ in range(100)], np.int32)
        models, ok = O.solve_pnp(X, noisy, K, samples)
        assert ok.all()
        med.append(np.median(np.abs(models - T).max(1)))
    assert med[0] > med[1] > med[2] and med[2] < 2e-3 and med[0] < 0.05


def test_epnp_reports_degenerate_samples():
    X, uv, K, _ = exact_pnp(n=40, seed=3)
    line = X.copy()
    line[:, 1] = 2.0 * line[:, 0] + 0.5                                # collinear object points: one axis of spread only
    line[:, 2] = -0.7 * line[:, 0] + 4.0
    same = np.repeat(X[:1], 40, axis=0)                               # one point forty times
    for pts in (line, same):
        models, ok = O.solve_pnp(pts, uv, K, np.arange(10, dtype=np.int32).reshape(2, 5))
        assert not ok.any() and not models.any()


def test_epnp_on_coplanar_points_takes_the_three_control_point_form():
    """cv::solvePnPRansac(..., SOLVEPNP_EPNP) returns a pose for a planar target (ReconstructionManager.cpp:227-228); the
    four-control-point form has no volume to work with there. Exact pixels of a plane -> the exact pose (no mirror solution:
    the perspective of the scene decides), from 5 points up to all of them; noisy pixels -> an error that shrinks with the
    sample size, as for a scene with volume; a plane in general position and the three axis-aligned planes (the flat axis is
    each of the eigenproblem's three positions)."""
    for noise, bounds in ((0.0, (1e-9, 1e-10, 1e-10)), (0.5, (0.05, 0.015, 3e-3)), (2.0, (0.2, 0.06, 0.012))):
        X, uv, K, T = SC.planar_pnp_case(n=300, seed=3, noise=noise)
        rng = np.random.default_rng(1)
        med = []
        for m in (5, 20, 299):
            samples = np.array([rng.choice(len(X), m, replace=False) for _ in range(100)], np.int32)
            models, ok = O.solve_pnp(X, uv, K, samples)
            assert ok.all()
            R = models[:, :9].reshape(-1, 3, 3)
            assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-12 and np.abs(np.linalg.det(R) - 1).max() < 1e-12
            med.append(np.median(np.abs(models - T).max(1)))
        assert all(a < b for a, b in zip(med, bounds)), (noise, med)
        if noise > 0:
            assert med[0] > med[1] > med[2]
    Xe, uve, K, T = exact_pnp(n=61, seed=5)                             # (the helper drops the scene's first point: 60 left)
    Rm, t = T[:9].reshape(3, 3), T[9:]
    for axis in range(3):
        X = Xe.copy()
        X[:, axis] = 0.25                                              # the plane x = c, y = c, z = c
        pc = X @ Rm.T + t
        uv = np.stack([K[0] * pc[:, 0] / pc[:, 2] + K[2], K[1] * pc[:, 1] / pc[:, 2] + K[3]], 1)
        models, ok = O.solve_pnp(X, uv, K, np.arange(60, dtype=np.int32).reshape(12, 5))
        assert ok.all() and np.abs(models - T).max() < 1e-8, axis