"""GPU: the two forms of RefineBA's graph construction (modules/sfm/reconstruction/BundleAdjuster.cpp:57-178) behind
eacham_ba_prepare — host loops (local windows) and device sorts + scans (eacham_amd/csrc/devprim.hpp, ba.hip) — must build
the SAME structure: every fp64 sum of the solver runs in the order of these lists, so the reduced system, the step and a
whole Levenberg-Marquardt run are compared BIT FOR BIT between a context forced to one form and a context forced to the
other (EACHAM_BA_PREPARE is read at eacham_ctx_create). Parity with the oracle is the business of tests/test_ba_gpu.py,
which runs on whichever form the problem size selects."""
import os

import numpy as np
import pytest

from eacham_amd import HipContext, ba, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["groups", "pairs"])
def ctx_pair(request):
    """(host-built, device-built) contexts under one form of the Schur stage: the landmark groups of csrc/ba_groups.hpp (what the
    device form builds by default) or the pair lists of rounds 1-4 (what the host form builds by default) — either form of the
    construction must reproduce either structure."""
    old = {k: os.environ.get(k) for k in ("EACHAM_BA_PREPARE", "EACHAM_BA_SCHUR")}
    try:
        os.environ["EACHAM_BA_SCHUR"] = request.param
        os.environ["EACHAM_BA_PREPARE"] = "host"
        host = HipContext(0)
        os.environ["EACHAM_BA_PREPARE"] = "device"
        dev = HipContext(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    yield host, dev
    host.close()
    dev.close()


def _arrays(seed, n_cams, n_lm, k, shuffle=False, **kw):
    sc = synth.make_scene(n_cams, n_lm, k, seed=seed, **kw)
    A = ba.BaArrays.from_scene(sc)
    A.obs_uv[::13] += 30.0
    if shuffle:  # the caller's observation order is arbitrary: within a landmark it must be kept (a stable sort)
        perm = np.random.default_rng(seed).permutation(len(A.obs_cam))
        A.obs_cam, A.obs_point, A.obs_uv = A.obs_cam[perm].copy(), A.obs_point[perm].copy(), A.obs_uv[perm].copy()
    return A


def _same_structure(ctx_pair, A):
    host, dev = ctx_pair
    a, b = ba.PreparedBA(host, A), ba.PreparedBA(dev, A)
    try:
        for name in ba.PreparedBA.STRUCTURE:
            x, y = a.structure(name), b.structure(name)
            assert x.shape == y.shape and np.array_equal(x, y), f"structure array {name} differs ({x.shape} vs {y.shape})"
    finally:
        a.close()
        b.close()


def _same_step(ctx_pair, A, lam=1e-3):
    host, dev = ctx_pair
    _same_structure(ctx_pair, A)
    a = ba.debug_step(host, A, lam)
    b = ba.debug_step(dev, A, lam)
    for name, x, y in zip(["S", "g", "delta_c", "delta_l", "error", "lin_change"], a, b):
        assert np.array_equal(np.asarray(x), np.asarray(y)), f"{name} differs between the host-built and the device-built structure"


def _same_run(ctx_pair, A, cfg=None):
    host, dev = ctx_pair
    cfg = cfg or ba.OptimizerConfig.refine_ba()
    a, b = ba.RefineBA(host, A, cfg), ba.RefineBA(dev, A, cfg)
    assert (a.outer_iterations, a.inner_iterations, a.status) == (b.outer_iterations, b.inner_iterations, b.status)
    assert a.initial_error == b.initial_error and a.final_error == b.final_error
    assert np.array_equal(a.trace, b.trace)
    assert np.array_equal(a.cam_T_wc, b.cam_T_wc) and np.array_equal(a.points, b.points) and np.array_equal(a.K, b.K)
    return a


@pytest.mark.parametrize("n_cams,n_lm,k,shuffle", [(5, 80, 3, False), (12, 300, 6, True), (31, 400, 6, False), (60, 600, 8, True),
                                                   (130, 2600, 6, True)])
def test_reduced_system_and_step_are_bit_identical(ctx_pair, n_cams, n_lm, k, shuffle):
    _same_step(ctx_pair, _arrays(7 + n_cams, n_cams, n_lm, k, shuffle=shuffle, pixel_noise=1.5))


def test_repeated_camera_unobserved_landmark_and_idle_camera(ctx_pair):
    A = _arrays(4, 6, 90, 3)
    # the same camera observes one landmark twice (two entries of its diagonal block per pair), an unobserved landmark
    # (an empty segment in the middle of lm_ptr), a camera nobody observes through (a diagonal block with no entry)
    A.obs_cam = np.concatenate([A.obs_cam, A.obs_cam[:5]]).astype(np.uint32)
    A.obs_point = np.concatenate([A.obs_point, A.obs_point[:5]]).astype(np.uint32)
    A.obs_uv = np.concatenate([A.obs_uv, A.obs_uv[:5] + 0.7])
    keep = A.obs_point != 17
    A.obs_cam, A.obs_point, A.obs_uv = A.obs_cam[keep].copy(), A.obs_point[keep].copy(), A.obs_uv[keep].copy()
    keep = A.obs_cam != 3
    A.obs_cam, A.obs_point, A.obs_uv = A.obs_cam[keep].copy(), A.obs_point[keep].copy(), A.obs_uv[keep].copy()
    _same_step(ctx_pair, A)
    _same_run(ctx_pair, A)


def test_whole_runs_are_bit_identical(ctx_pair):
    for A in [_arrays(21, 19, 1500, 7, shuffle=True, pixel_noise=1.0), _arrays(22, 130, 2600, 6, pixel_noise=1.0)]:
        out = _same_run(ctx_pair, A)
        assert out.outer_iterations >= 2 and out.final_error < out.initial_error
    _same_run(ctx_pair, _arrays(23, 40, 900, 6, pixel_noise=1.0), ba.OptimizerConfig("DogLeg", 30, 1e-5, 10.0, False))


def test_out_of_range_observation_is_refused_by_both(ctx_pair):
    from eacham_amd import EachamError, capi
    A = _arrays(5, 6, 90, 3)
    A.obs_point[11] = 90  # one past the last landmark
    for ctx in ctx_pair:
        with pytest.raises(EachamError) as e:
            ba.RefineBA(ctx, A, ba.OptimizerConfig.refine_ba())
        assert e.value.code == capi.ERR_INVALID


def test_metric_scene_is_bit_identical_and_reports_its_preparation(ctx_pair):
    """S200 (200 cameras / 50 000 landmarks / 500 000 observations): the size the device form is for."""
    host, dev = ctx_pair
    A = ba.BaArrays.from_scene(synth.make_scene(200, 50_000, 10, seed=12345))
    out = _same_run(ctx_pair, A)
    assert out.outer_iterations == 4
    infos = []
    for ctx in (host, dev):
        pb = ba.PreparedBA(ctx, A)
        infos.append(pb.plan_info())
        pb.close()
    assert infos[0]["panels"] == infos[1]["panels"] and infos[0]["tiles"] == infos[1]["tiles"] and infos[0]["levels"] == infos[1]["levels"]
    print("prepare_us host", infos[0]["prepare_us"], "device", infos[1]["prepare_us"])
