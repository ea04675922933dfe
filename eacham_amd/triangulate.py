"""Host-side mirror of the reference triangulation interface, on top of the C-ABI.

  triangulate_tracks(...)         <->  the per-point loop of TriangulateFrame
                                       (modules/sfm/reconstruction/Triangulator.cpp:248-283)
  TriangulatePointRansac(data,..) <->  Triangulator.cpp:96-186 (one track)
  reprojection_errors(...)        <->  CalcReprojectionError (ProjectionHelper.cpp:32-38)

`minTriAngle` is in radians here, as it is by the time the reference calls these functions
(SfmConfig.h:52-53 converts the configured degrees). Python is only the test/bench driver; the C++
adapter is include/eacham/TriangulatorHip.hpp.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import capi
from .matcher import HipContext

STATUS_RANSAC = 1  # TriangulatePointRansac returned true
STATUS_FULL = 2    # non-empty mask, every observation an inlier
STATUS_ADD = 3     # TriangulateFrame adds the point


@dataclass
class EstimatorData:
    """modules/sfm/reconstruction/Triangulator.h: one observation of a track."""
    point2d: np.ndarray    # pixel (x, y)
    transform: np.ndarray  # 4x4 world->camera
    K: np.ndarray          # 3x3 or (fx, fy, cx, cy)


def _K4(K) -> np.ndarray:
    K = np.asarray(K, dtype=np.float64)
    if K.shape == (3, 3):
        return np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]])
    return np.ascontiguousarray(K.reshape(4))


def triangulate_tracks(ctx: HipContext, transforms, track_ptr, obs_frame, obs_uv, K, max_repr_error: float,
                       min_tri_angle: float):
    """Returns (points n x 3, status n int32, masks n_obs uint8); see include/eacham_hip.h."""
    T = np.ascontiguousarray(transforms, dtype=np.float64).reshape(-1, 16)
    tp = np.ascontiguousarray(track_ptr, dtype=np.int32)
    of = np.ascontiguousarray(obs_frame, dtype=np.uint32)
    uv = np.ascontiguousarray(obs_uv, dtype=np.float64).reshape(-1, 2)
    n = tp.size - 1
    if n < 0 or of.size != uv.shape[0] or (n >= 0 and tp.size and int(tp[-1]) != of.size):
        raise ValueError("track_ptr / observation arrays disagree")
    K4 = _K4(K)
    points = np.zeros((max(n, 0), 3), dtype=np.float64)
    status = np.zeros(max(n, 0), dtype=np.int32)
    masks = np.zeros(of.size, dtype=np.uint8)
    ctx._check(ctx._L.eacham_triangulate_tracks(
        ctx.handle, T.ctypes.data, T.shape[0], n, tp.ctypes.data, of.ctypes.data, uv.ctypes.data, K4.ctypes.data,
        float(max_repr_error), float(min_tri_angle), points.ctypes.data, status.ctypes.data, masks.ctypes.data))
    return points, status, masks


def TriangulatePointRansac(ctx: HipContext, data: list, maxReprError: float, minTriAngle: float):
    """One track; returns (ok, point3d, inliers) as the reference's (return, point3d&, inliers&)."""
    m = len(data)
    if m == 0:
        return False, np.zeros(3), []
    T = np.stack([np.asarray(d.transform, dtype=np.float64).reshape(16) for d in data])
    uv = np.stack([np.asarray(d.point2d, dtype=np.float64).reshape(2) for d in data])
    pts, status, masks = triangulate_tracks(ctx, T, [0, m], np.arange(m), uv, data[0].K, maxReprError, minTriAngle)
    return bool(status[0] & STATUS_RANSAC), pts[0], [bool(x) for x in masks]


def reprojection_errors(ctx: HipContext, transforms, frame, points, uv, K) -> np.ndarray:
    T = np.ascontiguousarray(transforms, dtype=np.float64).reshape(-1, 16)
    fr = np.ascontiguousarray(frame, dtype=np.uint32)
    P = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    U = np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2)
    if not (fr.size == P.shape[0] == U.shape[0]):
        raise ValueError("frame / points / uv lengths disagree")
    K4 = _K4(K)
    err = np.zeros(fr.size, dtype=np.float32)
    ctx._check(ctx._L.eacham_reprojection_errors(ctx.handle, T.ctypes.data, T.shape[0], fr.size, fr.ctypes.data,
                                                 P.ctypes.data, U.ctypes.data, K4.ctypes.data, err.ctypes.data))
    return err


def two_view_points(ctx: HipContext, uv1, uv2, K, transforms, max_repr_error: float, min_tri_angle: float,
                    angle_strict: bool):
    """RecoverPoseTwoView's per-match loops (ReconstructionManager.cpp:118-143 / :162-186) for candidate
    relative poses. Returns (points nt x n x 3, keep nt x n uint8, counts nt)."""
    U1 = np.ascontiguousarray(uv1, dtype=np.float64).reshape(-1, 2)
    U2 = np.ascontiguousarray(uv2, dtype=np.float64).reshape(-1, 2)
    T = np.ascontiguousarray(transforms, dtype=np.float64).reshape(-1, 16)
    if U1.shape != U2.shape:
        raise ValueError("uv1 / uv2 lengths disagree")
    n, nt = U1.shape[0], T.shape[0]
    K4 = _K4(K)
    pts = np.zeros((nt, n, 3), dtype=np.float64)
    keep = np.zeros((nt, n), dtype=np.uint8)
    counts = np.zeros(nt, dtype=np.int32)
    ctx._check(ctx._L.eacham_two_view_points(ctx.handle, n, U1.ctypes.data, U2.ctypes.data, K4.ctypes.data, nt, T.ctypes.data,
                                             float(max_repr_error), float(min_tri_angle), int(bool(angle_strict)),
                                             pts.ctypes.data, keep.ctypes.data, counts.ctypes.data))
    return pts, keep, counts
