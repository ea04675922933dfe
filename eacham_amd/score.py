"""Host-side mirror of eacham_score_hypotheses (include/eacham_hip.h): batch scoring of essential-matrix,
homography and PnP pose candidates — the data-parallel part of the robust estimators RecoverPoseTwoView and
RecoverPosePnP call (modules/sfm/reconstruction/ReconstructionManager.cpp:57-61, :75, :227-228). Test / bench driver."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

KINDS = {"essential": capi.SCORE_ESSENTIAL, "homography": capi.SCORE_HOMOGRAPHY, "pnp": capi.SCORE_PNP}


def marshal(kind: str, a, b, models, K):
    k = KINDS[kind]
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 3 if kind == "pnp" else 2)
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1, 2)
    models = np.ascontiguousarray(models, dtype=np.float64).reshape(-1, 12 if kind == "pnp" else 9)
    if a.shape[0] != b.shape[0]:
        raise ValueError("point lists disagree")
    K4 = None if K is None else np.ascontiguousarray(K, dtype=np.float64).reshape(4)
    return k, a, b, models, K4


def score_hypotheses(ctx, kind: str, a, b, models, K=None, threshold: float = 16.0, want_errors: bool = True):
    """Returns (errors [n_models, n] float32 or None, inlier_counts [n_models] int32, medians [n_models] float32)."""
    k, a, b, models, K4 = marshal(kind, a, b, models, K)
    n, nm = a.shape[0], models.shape[0]
    err = np.zeros((nm, n), dtype=np.float32) if want_errors else None
    counts = np.zeros(nm, dtype=np.int32)
    med = np.zeros(nm, dtype=np.float32)
    vp = C.c_void_p
    ctx._check(capi.lib().eacham_score_hypotheses(
        ctx.handle, k, n, vp(a.ctypes.data), vp(b.ctypes.data), nm, vp(models.ctypes.data),
        vp(K4.ctypes.data) if K4 is not None else None, C.c_float(threshold), vp(err.ctypes.data) if want_errors else None,
        vp(counts.ctypes.data), vp(med.ctypes.data)))
    return err, counts, med


SOLVERS = {"homography4": (capi.SOLVE_HOMOGRAPHY4, 4, 1), "essential5": (capi.SOLVE_ESSENTIAL5, 5, 10)}


def solve_minimal(ctx, kind: str, a, b, samples, K=None):
    """eacham_solve_minimal: every minimal sample (rows of `samples`: 4 / 5 point indices) -> its model(s).
    Returns (models [n_samples, max_models, 9] float64, n_models [n_samples] int32); max_models = 1 / 10."""
    k, m, maxm = SOLVERS[kind]
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 2)
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1, 2)
    idx = np.ascontiguousarray(samples, dtype=np.int32).reshape(-1, m)
    if a.shape[0] != b.shape[0]:
        raise ValueError("point lists disagree")
    K4 = None if K is None else np.ascontiguousarray(K, dtype=np.float64).reshape(4)
    models = np.zeros((idx.shape[0], maxm, 9), dtype=np.float64)
    counts = np.zeros(idx.shape[0], dtype=np.int32)
    vp = C.c_void_p
    ctx._check(capi.lib().eacham_solve_minimal(ctx.handle, k, a.shape[0], vp(a.ctypes.data), vp(b.ctypes.data),
                                               vp(K4.ctypes.data) if K4 is not None else None, idx.shape[0], vp(idx.ctypes.data),
                                               vp(models.ctypes.data), vp(counts.ctypes.data)))
    return models, counts


def solve_pnp(ctx, object_points, image_points, K, samples):
    """eacham_solve_pnp: EPnP on every row of `samples` (>= 5 point indices per row) -> (models [n_samples, 12] = R | t, ok [n_samples])."""
    X = np.ascontiguousarray(object_points, dtype=np.float64).reshape(-1, 3)
    uv = np.ascontiguousarray(image_points, dtype=np.float64).reshape(-1, 2)
    idx = np.ascontiguousarray(samples, dtype=np.int32)
    if idx.ndim != 2 or X.shape[0] != uv.shape[0]:
        raise ValueError("samples must be [n_samples, sample_size]; point lists must agree")
    K4 = np.ascontiguousarray(K, dtype=np.float64).reshape(4)
    models = np.zeros((idx.shape[0], 12), dtype=np.float64)
    ok = np.zeros(idx.shape[0], dtype=np.int32)
    vp = C.c_void_p
    ctx._check(capi.lib().eacham_solve_pnp(ctx.handle, X.shape[0], vp(X.ctypes.data), vp(uv.ctypes.data), vp(K4.ctypes.data), idx.shape[1],
                                           idx.shape[0], vp(idx.ctypes.data), vp(models.ctypes.data), vp(ok.ctypes.data)))
    return models, ok
