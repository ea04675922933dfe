"""GPU: eacham_score_hypotheses (through the C-ABI) against the CPU oracle — errors, inlier counts and medians must be
bit-identical (float results of the same round-to-nearest operations in the same order)."""
import numpy as np
import pytest

from eacham_amd import capi, score, EachamError
import oracle_api as O
import score_cases as SC

pytestmark = pytest.mark.gpu


def _same(got, want):
    assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)), np.abs(got[0] - want[0]).max()
    assert np.array_equal(got[1], want[1])
    assert np.array_equal(got[2].view(np.uint32), want[2].view(np.uint32))


@pytest.mark.parametrize("n", [1, 2, 63, 700, 701, 4097])
def test_essential_parity(hip_ctx, n):
    c = SC.two_view_case(n=max(n, 8))
    uv1, uv2 = c["uv1"][:n], c["uv2"][:n]
    thr = (1.5 / c["K"][0]) ** 2
    _same(score.score_hypotheses(hip_ctx, "essential", uv1, uv2, c["E"], c["K"], thr), O.score_hypotheses("essential", uv1, uv2, c["E"], c["K"], thr))
    x = np.stack([(uv1[:, 0] - c["K"][2]) / c["K"][0], (uv1[:, 1] - c["K"][3]) / c["K"][1]], 1)
    y = np.stack([(uv2[:, 0] - c["K"][2]) / c["K"][0], (uv2[:, 1] - c["K"][3]) / c["K"][1]], 1)
    _same(score.score_hypotheses(hip_ctx, "essential", x, y, c["E"], None, thr), O.score_hypotheses("essential", x, y, c["E"], None, thr))


@pytest.mark.parametrize("n", [5, 500, 1024])
def test_homography_parity(hip_ctx, n):
    c = SC.two_view_case(n=n, planar=True, seed=13)
    _same(score.score_hypotheses(hip_ctx, "homography", c["uv1"], c["uv2"], c["H"], None, 16.0),
          O.score_hypotheses("homography", c["uv1"], c["uv2"], c["H"], None, 16.0))


def test_pnp_parity_at_the_reference_hypothesis_count(hip_ctx):
    """solvePnPRansac(..., 10000 iterations, 4 px): all 10 000 candidate poses in one call."""
    c = SC.pnp_case(n=800, n_models=10_000, seed=17)
    got = score.score_hypotheses(hip_ctx, "pnp", c["X"], c["uv"], c["models"], c["K"], 16.0)
    _same(got, O.score_hypotheses("pnp", c["X"], c["uv"], c["models"], c["K"], 16.0))
    assert got[1][0] >= got[1].max() - 3 and got[1][0] > 0.6 * 800   # the true pose ties with its tiny perturbations


def test_more_points_than_fit_the_lds_key_buffer(hip_ctx):
    """> 16384 correspondences: the median is selected from the error matrix in memory instead of LDS keys."""
    c = SC.pnp_case(n=20_001, n_models=5, seed=19)
    _same(score.score_hypotheses(hip_ctx, "pnp", c["X"], c["uv"], c["models"], c["K"], 16.0),
          O.score_hypotheses("pnp", c["X"], c["uv"], c["models"], c["K"], 16.0))
    e2, c2, m2 = score.score_hypotheses(hip_ctx, "pnp", c["X"], c["uv"], c["models"], c["K"], 16.0, want_errors=False)
    assert e2 is None and np.array_equal(c2, O.score_hypotheses("pnp", c["X"], c["uv"], c["models"], c["K"], 16.0)[1])


def test_empty_and_errors(hip_ctx):
    c = SC.pnp_case(n=50, n_models=3)
    err, cnt, med = score.score_hypotheses(hip_ctx, "pnp", c["X"][:0], c["uv"][:0], c["models"], c["K"], 16.0)
    assert err.shape == (3, 0) and not cnt.any() and np.isnan(med).all()
    err, cnt, med = score.score_hypotheses(hip_ctx, "pnp", c["X"], c["uv"], c["models"][:0], c["K"], 16.0)
    assert cnt.shape == (0,)
    with pytest.raises(EachamError) as e:
        score.score_hypotheses(hip_ctx, "pnp", c["X"], c["uv"], c["models"], None, 16.0)    # PnP needs K
    assert e.value.code == capi.ERR_INVALID
