"""CPU: the hypothesis-scoring oracle (oracle/score_oracle.c) against an independent numpy statement of the same
OpenCV callbacks, and against the geometry (the true model separates inliers from outliers)."""
import numpy as np
import pytest

import oracle_api as O
import score_cases as SC


def np_essential(uv1, uv2, E, K):
    x1 = np.stack([(uv1[:, 0] - K[2]) / K[0], (uv1[:, 1] - K[3]) / K[1], np.ones(len(uv1))], 1)
    x2 = np.stack([(uv2[:, 0] - K[2]) / K[0], (uv2[:, 1] - K[3]) / K[1], np.ones(len(uv2))], 1)
    out = []
    for M in E.reshape(-1, 3, 3):
        Ex1, Etx2 = x1 @ M.T, x2 @ M
        num = np.einsum("ij,ij->i", x2, Ex1) ** 2
        out.append(num / (Ex1[:, 0] ** 2 + Ex1[:, 1] ** 2 + Etx2[:, 0] ** 2 + Etx2[:, 1] ** 2))
    return np.array(out)


def np_homography(uv1, uv2, H):
    out = []
    for M in H.reshape(-1, 3, 3):
        p = np.concatenate([uv1, np.ones((len(uv1), 1))], 1) @ (M / 1.0).T
        w = uv1[:, 0] * M[2, 0] + uv1[:, 1] * M[2, 1] + 1.0
        out.append(((p[:, 0] / w - uv2[:, 0]) ** 2 + (p[:, 1] / w - uv2[:, 1]) ** 2))
    return np.array(out)


def np_pnp(X, uv, models, K):
    out = []
    for M in models:
        R, t = M[:9].reshape(3, 3), M[9:]
        pc = X @ R.T + t
        z = np.where(pc[:, 2] != 0, 1.0 / np.where(pc[:, 2] != 0, pc[:, 2], 1.0), 1.0)
        u, v = pc[:, 0] * z * K[0] + K[2], pc[:, 1] * z * K[1] + K[3]
        out.append((uv[:, 0] - u) ** 2 + (uv[:, 1] - v) ** 2)
    return np.array(out)


def np_median(e):
    s = np.sort(e.astype(np.float32), axis=1)
    n = e.shape[1]
    return s[:, n // 2] if n % 2 else ((s[:, n // 2 - 1] + s[:, n // 2]) * np.float32(0.5)).astype(np.float32)


@pytest.mark.parametrize("n", [700, 701])
def test_essential_scores(n):
    c = SC.two_view_case(n=n)
    thr = (1.5 / c["K"][0]) ** 2                       # findEssentialMat scales the pixel threshold by the focal length
    err, cnt, med = O.score_hypotheses("essential", c["uv1"], c["uv2"], c["E"], c["K"], thr)
    ref = np_essential(c["uv1"], c["uv2"], c["E"], c["K"])
    assert np.allclose(err, ref, rtol=2e-6, atol=0) and err.dtype == np.float32
    assert np.array_equal(cnt, (err <= np.float32(thr)).sum(1)) and np.array_equal(med, np_median(err))
    good = ~c["bad"]
    assert cnt[0] == cnt.max() and (err[0][good] <= thr).mean() > 0.95 and (err[0][c["bad"]] <= thr).mean() < 0.1
    assert med[0] <= 1.1 * med.min()                    # under LMedS the true model ties with its tiny perturbations, far ahead of the rest


def test_essential_without_K_takes_normalised_points():
    c = SC.two_view_case(n=300)
    K = c["K"]
    x1 = np.stack([(c["uv1"][:, 0] - K[2]) / K[0], (c["uv1"][:, 1] - K[3]) / K[1]], 1)
    x2 = np.stack([(c["uv2"][:, 0] - K[2]) / K[0], (c["uv2"][:, 1] - K[3]) / K[1]], 1)
    a = O.score_hypotheses("essential", x1, x2, c["E"], None, 1e-6)
    b = O.score_hypotheses("essential", c["uv1"], c["uv2"], c["E"], K, 1e-6)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("n", [500, 333])
def test_homography_scores(n):
    c = SC.two_view_case(n=n, planar=True, seed=13)
    err, cnt, med = O.score_hypotheses("homography", c["uv1"], c["uv2"], c["H"], None, 16.0)
    ref = np_homography(c["uv1"], c["uv2"], c["H"])
    ok = np.isfinite(ref) & (ref < 1e12)
    assert np.allclose(err[ok], ref[ok], rtol=2e-3, atol=2e-3)   # float arithmetic on pixel-sized numbers vs double
    assert np.array_equal(cnt, (err <= np.float32(16.0)).sum(1)) and np.array_equal(med, np_median(err))
    good = ~c["bad"]
    assert cnt[0] == cnt.max() and (err[0][good] <= 16.0).mean() > 0.95 and (err[0][c["bad"]] <= 16.0).mean() < 0.1


def test_pnp_scores():
    c = SC.pnp_case()
    err, cnt, med = O.score_hypotheses("pnp", c["X"], c["uv"], c["models"], c["K"], 16.0)   # 4 px, squared (:228)
    ref = np_pnp(c["X"], c["uv"], c["models"], c["K"])
    ok = ref < 1e10
    assert np.allclose(err[ok], ref[ok], rtol=1e-3, atol=1e-3)
    assert np.array_equal(cnt, (err <= np.float32(16.0)).sum(1)) and np.array_equal(med, np_median(err))
    good = ~c["bad"]
    good[0] = False                                     # the point at the camera centre
    assert cnt[0] == cnt.max() and (err[0][good] <= 16.0).mean() > 0.97
    assert np.isfinite(err[:, 0]).all()                 # z = 0 takes 1/z := 1 (cvProjectPoints2), never a NaN


def test_empty_inputs():
    c = SC.pnp_case(n=50, n_models=3)
    err, cnt, med = O.score_hypotheses("pnp", c["X"][:0], c["uv"][:0], c["models"], c["K"], 16.0)
    assert err.shape == (3, 0) and not cnt.any() and np.isnan(med).all()
    err, cnt, med = O.score_hypotheses("pnp", c["X"], c["uv"], c["models"][:0], c["K"], 16.0)
    assert err.shape == (0, 50) and cnt.shape == (0,)
