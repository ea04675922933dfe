"""GPU: triangulation through the C-ABI against the oracle."""
import os

import numpy as np
import pytest

import oracle_api as O
from eacham_amd import HipContext, capi, synth
from eacham_amd import triangulate as tri

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "tri_golden.npz")
MAX_ERR = 4.0
MIN_ANGLE = 3.0 * 3.141592 / 180.0


@pytest.fixture(scope="module")
def ctx():
    with HipContext(0) as c:
        yield c


def _compare(ctx, tr, max_err=MAX_ERR, min_angle=MIN_ANGLE):
    args = (tr["transforms"], tr["track_ptr"], tr["obs_frame"], tr["obs_uv"], tr["K"], max_err, min_angle)
    pts, status, masks = tri.triangulate_tracks(ctx, *args)
    opts, ostatus, omasks = O.tri_tracks(*args)
    # verdicts are threshold decisions on values that differ in the last bits between the two
    # builds (fma contraction); a flip is legitimate only when the deciding value sits on the
    # threshold, which seeded data does not produce: demand exact equality and report otherwise.
    diff = np.nonzero(status != ostatus)[0]
    assert diff.size == 0, f"verdicts differ on tracks {diff[:8]}"
    assert np.array_equal(masks, omasks)
    # the select kernel re-derives the winning pair's mask: it must agree with the count the pair kernel decided on
    # (status bit 1 = every observation an inlier), for tracks of any length
    ptr = np.asarray(tr["track_ptr"])
    full = np.array([masks[a:b].all() if b > a else False for a, b in zip(ptr[:-1], ptr[1:])])
    assert np.array_equal((status & 2) != 0, full), "mask and status disagree"
    fin = np.isfinite(opts).all(1)
    assert np.array_equal(np.isfinite(pts).all(1), fin)
    assert np.allclose(pts[fin], opts[fin], rtol=1e-9, atol=1e-11)
    return pts, status, masks


def test_golden(ctx):
    g = np.load(GOLD)
    pts, status, masks = tri.triangulate_tracks(ctx, g["transforms"], g["track_ptr"], g["obs_frame"], g["obs_uv"], g["K"],
                                                float(g["max_err"]), float(g["min_angle"]))
    assert np.array_equal(status, g["status"]) and np.array_equal(masks, g["masks"])
    fin = np.isfinite(g["points"]).all(1)
    assert np.allclose(pts[fin], g["points"][fin], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("seed,n_cams,n_lm,k", [(1, 12, 300, 5), (2, 40, 3000, 10), (3, 200, 20000, 10)])
def test_matches_oracle(ctx, seed, n_cams, n_lm, k):
    sc = synth.make_scene(n_cams, n_lm, k, seed=seed, pixel_noise=1.0)
    _, status, _ = _compare(ctx, synth.make_tracks(sc, seed=seed, min_obs=1, outlier_frac=0.2))
    assert set(np.unique(status)) == {0, 1, 2, 3}


def test_noisy_poses_and_tight_thresholds(ctx):
    sc = synth.make_scene(30, 2000, 8, seed=4, pixel_noise=2.0)
    tr = synth.make_tracks(sc, seed=4, min_obs=2, outlier_frac=0.5, use_true_poses=False)
    _compare(ctx, tr, max_err=8.0, min_angle=2.0 * 3.141592 / 180.0)  # the "init" block of SfmConfig.json:24-25


def test_long_tracks_have_no_limit(ctx):
    """The reference's TriangulatePointRansac takes any number of observations (Triangulator.cpp:96-186: the observer
    list of a map point of a long sequence): 64 (2016 pairs per track) and 100 observations (4950 pairs), with outliers,
    verdicts / masks identical to the oracle. (Until round 3 a 64-bit mask per pair capped a track at 64.)"""
    sc = synth.make_scene(64, 50, 64, seed=5, pixel_noise=0.5)
    _compare(ctx, synth.make_tracks(sc, seed=5, min_obs=64, outlier_frac=0.0))
    sc = synth.make_scene(100, 40, 100, seed=6, pixel_noise=0.5)
    tr = synth.make_tracks(sc, seed=6, min_obs=100, outlier_frac=0.5)
    assert np.diff(tr["track_ptr"]).max() == 100
    _compare(ctx, tr)
    mixed = synth.make_tracks(sc, seed=7, min_obs=2, outlier_frac=0.3)   # 2 .. 100 observations in one batch
    assert np.diff(mixed["track_ptr"]).max() > 64
    _compare(ctx, mixed)


def test_empty_short_and_invalid(ctx):
    K = np.array([960.0, 960.0, 400.0, 400.0])
    pts, status, masks = tri.triangulate_tracks(ctx, np.zeros((0, 16)), [0], [], np.zeros((0, 2)), K, MAX_ERR, MIN_ANGLE)
    assert pts.shape == (0, 3) and status.size == 0 and masks.size == 0
    T = np.eye(4).reshape(1, 16)
    pts, status, masks = tri.triangulate_tracks(ctx, T, [0, 0, 1], [0], [[1.0, 2.0]], K, MAX_ERR, MIN_ANGLE)  # 0 and 1 observations
    assert status.tolist() == [0, 0] and not pts.any() and masks.tolist() == [0]
    with pytest.raises(capi.EachamError) as e:  # frame index out of range must not reach the kernel
        tri.triangulate_tracks(ctx, T, [0, 2], [0, 7], np.zeros((2, 2)), K, MAX_ERR, MIN_ANGLE)
    assert e.value.code == capi.ERR_INVALID


def test_single_track_mirror(ctx):
    sc = synth.make_scene(12, 60, 5, seed=6, pixel_noise=1.0)
    tr = synth.make_tracks(sc, seed=6, min_obs=3, outlier_frac=0.0)
    o, e = tr["track_ptr"][0], tr["track_ptr"][1]
    data = [tri.EstimatorData(tr["obs_uv"][i], tr["transforms"][tr["obs_frame"][i]].reshape(4, 4), tr["K"]) for i in range(o, e)]
    ok, X, inl = tri.TriangulatePointRansac(ctx, data, MAX_ERR, MIN_ANGLE)
    opts, ostatus, omasks = O.tri_tracks(tr["transforms"], tr["track_ptr"][:2], tr["obs_frame"][:e], tr["obs_uv"][:e], tr["K"], MAX_ERR, MIN_ANGLE)
    assert ok == bool(ostatus[0] & 1) and inl == [bool(x) for x in omasks] and np.allclose(X, opts[0], rtol=1e-9)


def test_reprojection_errors(ctx):
    sc = synth.make_scene(20, 5000, 6, seed=8, pixel_noise=1.0)
    tr = synth.make_tracks(sc, seed=8)
    lm = np.repeat(tr["landmark"], np.diff(tr["track_ptr"]))
    X = sc["points_init"][lm]
    a = tri.reprojection_errors(ctx, tr["transforms"], tr["obs_frame"], X, tr["obs_uv"], tr["K"])
    b = O.reprojection_errors(tr["transforms"], tr["obs_frame"], X, tr["obs_uv"], tr["K"])
    assert np.allclose(a, b, rtol=2e-7, atol=1e-6)
    assert (np.abs(a - b) > 0).mean() < 0.05  # float rounding of an fp64 value: nearly always identical


def test_two_view_points(ctx):
    from test_tri_oracle import _two_view_case
    uv1, uv2, K, Ts = _two_view_case(seed=5, n=2000)
    for strict in (True, False):
        pts, keep, counts = tri.two_view_points(ctx, uv1, uv2, K, Ts, MAX_ERR, MIN_ANGLE, strict)
        opts, okeep, ocounts = O.two_view_points(uv1, uv2, K, Ts, MAX_ERR, MIN_ANGLE, strict)
        assert np.array_equal(keep, okeep) and np.array_equal(counts, ocounts)
        fin = np.isfinite(opts).all(2)
        assert np.allclose(pts[fin], opts[fin], rtol=1e-9, atol=1e-11)
        assert counts.argmax() == 2 and counts[2] > 1500
    pts, keep, counts = tri.two_view_points(ctx, np.zeros((0, 2)), np.zeros((0, 2)), K, Ts, MAX_ERR, MIN_ANGLE, True)
    assert pts.shape == (4, 0, 3) and counts.tolist() == [0, 0, 0, 0]
