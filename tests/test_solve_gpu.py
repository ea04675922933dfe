"""GPU: eacham_solve_minimal (eacham_amd/csrc/solve.hip) — the minimal solvers of cv::findHomography / cv::findEssentialMat
(ReconstructionManager.cpp:75, :57-61) for caller-supplied sample indices — against oracle/solve_oracle.c, BIT FOR BIT
(both sides use only + - * / sqrt and no fused multiply-adds), and end to end with eacham_score_hypotheses as an LMedS /
RANSAC loop whose sampling is this test's own (OpenCV's RNG stream is not reproducible: parity unpinned)."""
import numpy as np
import pytest

from eacham_amd import capi, score, EachamError
import oracle_api as O
import score_cases as SC

pytestmark = pytest.mark.gpu


def draw(rng, n, m, count):
    return np.array([rng.choice(n, m, replace=False) for _ in range(count)], np.int32)


def test_homography4_bit_identical_and_the_lmeds_loop_finds_the_plane(hip_ctx):
    c = SC.two_view_case(n=600, seed=13, outliers=0.3, planar=True)
    rng = np.random.default_rng(13)
    samples = draw(rng, 600, 4, 100)                      # findHomography(..., LMEDS, 4.0, mask, 100, 0.999): 100 iterations
    models, counts = score.solve_minimal(hip_ctx, "homography4", c["uv1"], c["uv2"], samples)
    want, wcounts = O.solve_minimal("homography4", c["uv1"], c["uv2"], samples)
    assert np.array_equal(counts, wcounts) and np.array_equal(models, want)
    # the LMedS choice over those models: the winner is the plane's homography up to the pixel noise
    _, inl, med = score.score_hypotheses(hip_ctx, "homography", c["uv1"], c["uv2"], models[:, 0], threshold=16.0, want_errors=False)
    best = models[int(np.argmin(med)), 0].reshape(3, 3)
    truth = c["H"][0].reshape(3, 3)
    p = np.c_[c["uv1"], np.ones(600)] @ best.T
    q = np.c_[c["uv1"], np.ones(600)] @ truth.T
    good = ~c["bad"]
    assert np.median(np.linalg.norm(p[good, :2] / p[good, 2:] - q[good, :2] / q[good, 2:], axis=1)) < 3.0
    assert inl.max() > 0.5 * good.sum()


def test_essential5_bit_identical_at_the_reference_iteration_count(hip_ctx):
    c = SC.two_view_case(n=700, seed=21, outliers=0.25)
    rng = np.random.default_rng(21)
    samples = draw(rng, 700, 5, 1000)                     # findEssentialMat(..., LMEDS, 0.99, 4.0, 1000, mask): 1000 iterations
    models, counts = score.solve_minimal(hip_ctx, "essential5", c["uv1"], c["uv2"], samples, c["K"])
    want, wcounts = O.solve_minimal("essential5", c["uv1"], c["uv2"], samples, c["K"])
    assert np.array_equal(counts, wcounts) and (counts % 2 == 0).all() and counts.max() <= 10
    assert np.array_equal(models, want)
    # every candidate of every sample scored in one call; the LMedS winner agrees with the true E on the inliers
    cand = np.concatenate([m[:n] for m, n in zip(models, counts)])
    _, inl, med = score.score_hypotheses(hip_ctx, "essential", c["uv1"], c["uv2"], cand, c["K"], threshold=16.0 / c["K"][0] ** 2,
                                         want_errors=False)
    best = cand[int(np.argmin(med))].reshape(3, 3)
    Et = c["E"][0].reshape(3, 3) / np.linalg.norm(c["E"][0])
    assert min(np.abs(best - Et).max(), np.abs(best + Et).max()) < 0.05
    assert len(cand) > 2000


def test_solver_arguments(hip_ctx):
    pts = np.zeros((6, 2))
    with pytest.raises(EachamError) as e:
        score.solve_minimal(hip_ctx, "homography4", pts, pts, np.array([[0, 1, 2, 6]], np.int32))   # index out of range
    assert e.value.code == capi.ERR_INVALID
    m, k = score.solve_minimal(hip_ctx, "essential5", pts, pts, np.array([[0, 1, 2, 3, 4]], np.int32))  # degenerate: no model
    assert k[0] == 0 and not m.any()
    m, k = score.solve_minimal(hip_ctx, "homography4", pts, pts, np.zeros((0, 4), np.int32))
    assert m.shape == (0, 1, 9)


def test_epnp_bit_identical_at_the_reference_iteration_count_and_the_ransac_loop_finds_the_pose(hip_ctx):
    """cv::solvePnPRansac(pts3d, pts2d, K, dist, rvec, t, false, 10000, 4.0f, 0.999f, inliers, SOLVEPNP_EPNP) (:227-228) as three
    launches: EPnP on 10 000 five-point samples, every model against every point (4 px), EPnP on the winner's inliers."""
    c = SC.pnp_case(n=500, seed=11, outliers=0.3)
    X, uv, K, T = c["X"][1:], c["uv"][1:], c["K"], c["models"][0]
    rng = np.random.default_rng(11)
    samples = draw(rng, len(X), 5, 10000)
    models, ok = score.solve_pnp(hip_ctx, X, uv, K, samples)
    want, wok = O.solve_pnp(X, uv, K, samples)
    assert np.array_equal(ok, wok) and np.array_equal(models, want) and ok.mean() > 0.99
    err, inl, _ = score.score_hypotheses(hip_ctx, "pnp", X, uv, models, K, threshold=16.0)
    best = int(np.argmax(inl))
    good = ~c["bad"][1:]
    assert inl[best] > 0.9 * good.sum()
    inliers = np.flatnonzero(err[best] <= 16.0).astype(np.int32)[None, :]
    refit, rok = score.solve_pnp(hip_ctx, X, uv, K, inliers)                       # the all-inlier refit: one row of ~350 indices
    wrefit, _ = O.solve_pnp(X, uv, K, inliers)
    assert rok[0] == 1 and np.array_equal(refit, wrefit)
    assert np.abs(refit[0] - T).max() < 5e-3 and np.abs(refit[0] - T).max() < np.abs(models[best] - T).max()
    # sample sizes on both sides of the one-thread / one-wave split (64) and not a multiple of the wave
    # (batches of <= 512 samples take the one-wave-per-sample kernel whatever their size, larger ones of <= 64 points the one-thread kernel)
    for m, count in ((12, 40), (64, 3), (65, 3), (200, 2), (5, 256), (5, 513), (30, 600)):
        rows = draw(rng, len(X), m, count)
        got, gok = score.solve_pnp(hip_ctx, X, uv, K, rows)
        want, wok = O.solve_pnp(X, uv, K, rows)
        assert np.array_equal(gok, wok) and np.array_equal(got, want), m


def test_epnp_on_coplanar_points_bit_identical_at_three_noise_levels(hip_ctx):
    """A planar target (cv::solvePnPRansac(..., SOLVEPNP_EPNP) returns a pose for it, ReconstructionManager.cpp:227-228): the
    three-control-point form, same bits as oracle/solve_oracle.c at every sample size class (a lane per start for the RANSAC
    loop's five-point samples, a wave per sample, the all-inlier refit), exact pose from exact pixels, and planar and spatial
    samples mixed in one batch (the branch is taken per sample)."""
    rng = np.random.default_rng(5)
    for noise, bound in ((0.0, 1e-9), (0.5, 0.05), (2.0, 0.2)):
        X, uv, K, T = SC.planar_pnp_case(n=300, seed=3, noise=noise)
        for m, count in ((5, 2000), (5, 300), (20, 40), (64, 3), (65, 3), (299, 2)):
            rows = draw(rng, len(X), m, count)
            got, gok = score.solve_pnp(hip_ctx, X, uv, K, rows)
            want, wok = O.solve_pnp(X, uv, K, rows)
            assert np.array_equal(gok, wok) and gok.all() and np.array_equal(got, want), (noise, m)
            if m == 5:
                assert np.median(np.abs(got - T).max(1)) < bound
            if m == 299:
                assert np.abs(got - T).max() < max(bound / 10, 1e-9)
    # a batch whose samples are planar or not: the first 150 object points on the plane, the rest off it
    X, uv, K, T = SC.planar_pnp_case(n=300, seed=3, noise=0.5)
    c = SC.pnp_case(n=300, seed=3, outliers=0.0)
    X2, uv2 = X.copy(), uv.copy()
    X2[150:], uv2[150:] = c["X"][1:][150:], c["uv"][1:][150:]
    rows = np.concatenate([draw(rng, 150, 5, 300), 150 + draw(rng, 149, 5, 300), draw(rng, 299, 5, 300)])
    got, gok = score.solve_pnp(hip_ctx, X2, uv2, K, rows)
    want, wok = O.solve_pnp(X2, uv2, K, rows)
    assert np.array_equal(gok, wok) and np.array_equal(got, want)


def test_epnp_arguments(hip_ctx):
    X, uv, K = np.zeros((8, 3)), np.zeros((8, 2)), np.array([500.0, 500.0, 320.0, 240.0])
    with pytest.raises(EachamError) as e:
        score.solve_pnp(hip_ctx, X, uv, K, np.array([[0, 1, 2, 3]], np.int32))                 # EPnP needs five points
    assert e.value.code == capi.ERR_INVALID
    with pytest.raises(EachamError):
        score.solve_pnp(hip_ctx, X, uv, K, np.array([[0, 1, 2, 3, 8]], np.int32))              # index out of range
    m, ok = score.solve_pnp(hip_ctx, X, uv, K, np.array([[0, 1, 2, 3, 4]], np.int32))          # coincident points: degenerate
    assert ok[0] == 0 and not m.any()
    X[:, 0] = np.arange(8.0)                                                                   # collinear points: degenerate
    m, ok = score.solve_pnp(hip_ctx, X, uv, K, np.array([[0, 1, 2, 3, 4], [3, 4, 5, 6, 7]], np.int32))
    assert not ok.any() and not m.any()
    m, ok = score.solve_pnp(hip_ctx, X, uv, K, np.zeros((0, 5), np.int32))
    assert m.shape == (0, 12)
