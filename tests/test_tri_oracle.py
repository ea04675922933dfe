"""CPU: the triangulation oracle against a literal numpy/LAPACK statement of the reference
(Triangulator.cpp:21-186) and against the committed fixture."""
import os

import numpy as np

import np_reference as R
import oracle_api as O
from eacham_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "tri_golden.npz")
MAX_ERR = 4.0
MIN_ANGLE = 3.0 * 3.141592 / 180.0


def _tracks(seed=3, n_cams=16, n_lm=120, k=6, **kw):
    sc = synth.make_scene(n_cams, n_lm, k, seed=seed, pixel_noise=1.0)
    return synth.make_tracks(sc, seed=seed, min_obs=1, **kw)


def test_null_vector_matches_lapack_svd():
    tr = _tracks()
    T, tp, fr, uv, K = tr["transforms"], tr["track_ptr"], tr["obs_frame"], tr["obs_uv"], tr["K"]
    checked = 0
    for t in range(tp.size - 1):
        o, m = tp[t], tp[t + 1] - tp[t]
        if m < 2:
            continue
        a = O.tri_point(T[fr[o]], T[fr[o + 1]], uv[o], uv[o + 1], K)
        b = R.tri_point(T[fr[o]], T[fr[o + 1]], uv[o], uv[o + 1], K)
        assert np.allclose(a, b, rtol=1e-8, atol=1e-10), (t, a, b)
        assert np.isclose(O.tri_angle(T[fr[o]], T[fr[o + 1]], a), R.tri_angle(T[fr[o]], T[fr[o + 1]], a), rtol=1e-10, atol=1e-12)
        checked += 1
    assert checked > 50


def test_degenerate_pair_does_not_crash():
    T = np.eye(4)
    p = O.tri_point(T, T, [400.0, 400.0], [400.0, 400.0], [960.0, 960.0, 400.0, 400.0])  # zero baseline
    assert p.shape == (3,)  # inf / nan allowed: the reference would produce the same kind of value
    assert O.tri_angle(T, T, np.zeros(3)) == 0.0  # ray of zero length -> `false` -> 0 (:27-35)


def test_ransac_matches_literal_restatement():
    tr = _tracks(seed=5, outlier_frac=0.3)
    T, tp, fr, uv, K = tr["transforms"], tr["track_ptr"], tr["obs_frame"], tr["obs_uv"], tr["K"]
    pts, status, masks = O.tri_tracks(T, tp, fr, uv, K, MAX_ERR, MIN_ANGLE)
    seen = set()
    for t in range(tp.size - 1):
        o, m = tp[t], tp[t + 1] - tp[t]
        ok, X, mask = R.tri_ransac([T[f] for f in fr[o:o + m]], list(uv[o:o + m]), K, MAX_ERR, MIN_ANGLE)
        full = len(mask) > 0 and sum(mask) == len(mask)
        assert (status[t] & 1) == int(ok), t
        assert ((status[t] >> 1) & 1) == int(full), t
        assert masks[o:o + m].tolist() == ([int(x) for x in mask] if mask else [0] * m), t
        if m >= 2:
            assert np.allclose(pts[t], X, rtol=1e-7, atol=1e-9), t
        seen.add(int(status[t]))
    assert seen == {0, 1, 2, 3}  # every verdict occurs, incl. the world-z quirk (status 2)


def test_last_pair_point_quirk():
    """The point returned for >= 3 observations is the LAST pair's triangulation (:136-141)."""
    tr = _tracks(seed=9, outlier_frac=0.0)
    T, tp, fr, uv, K = tr["transforms"], tr["track_ptr"], tr["obs_frame"], tr["obs_uv"], tr["K"]
    pts, _, _ = O.tri_tracks(T, tp, fr, uv, K, MAX_ERR, MIN_ANGLE)
    t = int(np.nonzero(np.diff(tp) >= 3)[0][0])
    e = tp[t + 1]
    last = O.tri_point(T[fr[e - 2]], T[fr[e - 1]], uv[e - 2], uv[e - 1], K)
    assert np.array_equal(pts[t], last)


def test_reprojection_errors():
    tr = _tracks(seed=11)
    n = tr["obs_frame"].size
    X = synth.rng_uniform(1, 2, (n, 3)) * 2 - 1
    err = O.reprojection_errors(tr["transforms"], tr["obs_frame"], X, tr["obs_uv"], tr["K"])
    T = tr["transforms"].reshape(-1, 4, 4)[tr["obs_frame"]]
    pc = np.einsum("nij,nj->ni", T[:, :3, :3], X) + T[:, :3, 3]
    K = tr["K"]
    u = K[0] * pc[:, 0] / pc[:, 2] + K[2]
    v = K[1] * pc[:, 1] / pc[:, 2] + K[3]
    ref = np.sqrt((tr["obs_uv"][:, 0] - u) ** 2 + (tr["obs_uv"][:, 1] - v) ** 2).astype(np.float32)
    assert np.allclose(err, ref, rtol=1e-6)


def test_golden_fixture():
    g = np.load(GOLD)
    pts, status, masks = O.tri_tracks(g["transforms"], g["track_ptr"], g["obs_frame"], g["obs_uv"], g["K"],
                                      float(g["max_err"]), float(g["min_angle"]))
    assert np.array_equal(status, g["status"]) and np.array_equal(masks, g["masks"])
    fin = np.isfinite(g["points"]).all(1)
    assert np.allclose(pts[fin], g["points"][fin], rtol=1e-9, atol=1e-12)
    assert set(np.unique(status)) == {0, 1, 2, 3}


def _two_view_case(seed=2, n=300):
    """Two cameras of a scene; camera 1 = identity: relative pose T21 = T2 T1^-1, plus wrong candidates
    (as cv::decomposeHomographyMat returns up to four)."""
    sc = synth.make_scene(8, n, 8, seed=seed, pixel_noise=1.0)
    T1, T2 = sc["T_true"][2], sc["T_true"][4]
    rel = T2 @ np.linalg.inv(T1)
    uv = sc["obs_uv"].reshape(n, 8, 2)
    uv1, uv2 = uv[:, 2].copy(), uv[:, 4].copy()
    uv2[::9] += 25.0                                   # wrong matches
    flip = rel.copy(); flip[:3, 3] *= -1               # the mirrored solution: points behind the cameras
    twist = rel.copy(); twist[:3, :3] = synth.so3_exp(np.array([0.0, 0.2, 0.0])) @ rel[:3, :3]
    scaled = rel.copy(); scaled[:3, 3] *= 0.01         # the scale of t is free: the same structure, 100x smaller
    return uv1, uv2, sc["K"], np.stack([flip, twist, rel, scaled])


def test_two_view_points_match_literal_restatement():
    uv1, uv2, K, Ts = _two_view_case()
    for strict in (True, False):
        pts, keep, counts = O.two_view_points(uv1, uv2, K, Ts, MAX_ERR, MIN_ANGLE, strict)
        for k in range(len(Ts)):
            rp, rk = R.two_view(uv1, uv2, K, Ts[k], MAX_ERR, MIN_ANGLE, strict)
            assert np.array_equal(keep[k].astype(bool), rk), (strict, k)
            fin = np.isfinite(rp).all(1)
            assert np.allclose(pts[k][fin], rp[fin], rtol=1e-7, atol=1e-9)
            assert counts[k] == rk.sum()
        assert counts.argmax() == 2 and counts[2] > 200 and counts[0] == 0 and counts[1] < 30 and counts[3] == counts[2]
