"""numpy front-end of the C oracle (oracle/*.c). Test infrastructure only."""
from __future__ import annotations

import ctypes as C

import numpy as np

import oracle


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def match_directed(A, B, ratio=0.8, force_f32=False):
    L = oracle.lib()
    A, B = _f32(A), _f32(B)
    n1, n2 = A.shape[0], B.shape[0]
    dim = A.shape[1] if A.ndim == 2 and A.shape[1] else (B.shape[1] if B.ndim == 2 else 0)
    q = np.empty(max(n1, 1), dtype=np.uint32)
    t = np.empty(max(n1, 1), dtype=np.uint32)
    L.oracle_match_directed.restype = C.c_int
    cnt = L.oracle_match_directed(C.c_void_p(A.ctypes.data), n1, C.c_void_p(B.ctypes.data), n2, dim,
                                  C.c_double(ratio), int(force_f32), C.c_void_p(q.ctypes.data),
                                  C.c_void_p(t.ctypes.data))
    return q[:cnt].copy(), t[:cnt].copy()


def knn2(A, B, force_f32=False):
    L = oracle.lib()
    A, B = _f32(A), _f32(B)
    n1, n2, dim = A.shape[0], B.shape[0], A.shape[1]
    idx = np.empty(max(n1, 1), dtype=np.int32)
    d0 = np.empty(max(n1, 1), dtype=np.float32)
    d1 = np.empty(max(n1, 1), dtype=np.float32)
    L.oracle_knn2.restype = None
    L.oracle_knn2(C.c_void_p(A.ctypes.data), n1, C.c_void_p(B.ctypes.data), n2, dim, int(force_f32),
                  C.c_void_p(idx.ctypes.data), C.c_void_p(d0.ctypes.data), C.c_void_p(d1.ctypes.data))
    return idx[:n1], d0[:n1], d1[:n1]


def match_mutual(A, B, ratio=0.8, min_dir=30, min_mutual=30, force_f32=False):
    L = oracle.lib()
    A, B = _f32(A), _f32(B)
    n1, n2 = A.shape[0], B.shape[0]
    dim = A.shape[1] if A.shape[1] else B.shape[1]
    q = np.empty(max(n1, 1), dtype=np.uint32)
    t = np.empty(max(n1, 1), dtype=np.uint32)
    stats = np.zeros(4, dtype=np.int32)
    L.oracle_match_mutual.restype = C.c_int
    cnt = L.oracle_match_mutual(C.c_void_p(A.ctypes.data), n1, C.c_void_p(B.ctypes.data), n2, dim,
                                C.c_double(ratio), min_dir, min_mutual, int(force_f32),
                                C.c_void_p(q.ctypes.data), C.c_void_p(t.ctypes.data),
                                C.c_void_p(stats.ctypes.data))
    return q[:cnt].copy(), t[:cnt].copy(), stats


def match_all_pairs(descs, pairs, ratio=0.8, min_dir=30, min_mutual=30, nthreads=0, force_f32=0):
    """Returns (counts, offsets, q, t, stats, threads_used) in the CSR form of the C-ABI."""
    L = oracle.lib()
    descs = [_f32(d) for d in descs]
    dim = max((d.shape[1] for d in descs if d.ndim == 2), default=0)
    n = np.array([d.shape[0] for d in descs], dtype=np.int32)
    ptrs = (C.c_void_p * len(descs))(*[d.ctypes.data for d in descs])
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    npairs = pairs.shape[0]
    stride = int(max(n.max() if len(n) else 1, 1))
    counts = np.zeros(npairs, dtype=np.int32)
    matches = np.zeros((npairs, stride, 2), dtype=np.uint32)
    stats = np.zeros((npairs, 4), dtype=np.int32)
    L.oracle_match_all_pairs.restype = C.c_int
    used = L.oracle_match_all_pairs(ptrs, C.c_void_p(n.ctypes.data), dim, C.c_void_p(pairs.ctypes.data),
                                    npairs, C.c_double(ratio), min_dir, min_mutual, nthreads,
                                    C.c_void_p(counts.ctypes.data), C.c_void_p(matches.ctypes.data),
                                    stride, C.c_void_p(stats.ctypes.data), int(force_f32))
    offsets = np.zeros(npairs + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    q = np.concatenate([matches[p, :counts[p], 0] for p in range(npairs)]) if npairs else np.zeros(0, np.uint32)
    t = np.concatenate([matches[p, :counts[p], 1] for p in range(npairs)]) if npairs else np.zeros(0, np.uint32)
    return counts, offsets, q.astype(np.uint32), t.astype(np.uint32), stats, used


# ---- bundle adjustment --------------------------------------------------------------------------
def ba_project(T_wc, point, K5, uv):
    L = oracle.lib()
    T = np.ascontiguousarray(T_wc, np.float64).reshape(16)
    pt = np.ascontiguousarray(point, np.float64)
    K = np.ascontiguousarray(K5, np.float64)
    m = np.ascontiguousarray(uv, np.float64)
    r = np.zeros(2); Jp = np.zeros((2, 6)); Jl = np.zeros((2, 3)); Jk = np.zeros((2, 5))
    L.oracle_ba_project.restype = C.c_int
    ok = L.oracle_ba_project(*[C.c_void_p(a.ctypes.data) for a in (T, pt, K, m, r, Jp, Jl, Jk)])
    return ok, r, Jp, Jl, Jk


def ba_pose_retract(T_wc, xi):
    L = oracle.lib()
    T = np.ascontiguousarray(T_wc, np.float64).reshape(16)
    x = np.ascontiguousarray(xi, np.float64)
    out = np.zeros(16)
    L.oracle_ba_pose_retract.restype = None
    L.oracle_ba_pose_retract(C.c_void_p(T.ctypes.data), C.c_void_p(x.ctypes.data), C.c_void_p(out.ctypes.data))
    return out.reshape(4, 4)


def ba_pose_local(T_wc, T_prior_wc):
    L = oracle.lib()
    a = np.ascontiguousarray(T_wc, np.float64).reshape(16)
    b = np.ascontiguousarray(T_prior_wc, np.float64).reshape(16)
    out = np.zeros(6)
    L.oracle_ba_pose_local.restype = None
    L.oracle_ba_pose_local(C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(out.ctypes.data))
    return out


def ba_error(arrays):
    L = oracle.lib()
    L.oracle_ba_error.restype = C.c_double
    prob = arrays.c_problem()
    return L.oracle_ba_error(C.byref(prob))


def ba_step(arrays, lam, mode=0):
    """(S, g, delta_cams, delta_points, error, lin_change, ok) of one damped step (mode 1 = dense solve)."""
    L = oracle.lib()
    prob = arrays.c_problem()
    n = 6 * prob.n_cams + 5
    S = np.zeros((n, n)); g = np.zeros(n); dc = np.zeros(n); dl = np.zeros((prob.n_points, 3))
    err = C.c_double(0); lin = C.c_double(0)
    L.oracle_ba_step.restype = C.c_int
    rc = L.oracle_ba_step(C.byref(prob), C.c_double(lam), int(mode), C.c_void_p(S.ctypes.data), C.c_void_p(g.ctypes.data),
                          C.c_void_p(dc.ctypes.data), C.c_void_p(dl.ctypes.data), C.byref(err), C.byref(lin))
    return S, g, dc, dl, err.value, lin.value, rc == 0


def ba_solve(arrays, cfg, min_landmarks=50, trace_cap=1024, nthreads=0, lm_factor="reset"):
    """cfg.usePreconditioner: the oracle solves every damped system with the PCG + block-Jacobi the reference can
    configure (BundleAdjuster.cpp:192-200) instead of the direct solve; `out.reserved` = PCG iterations in total."""
    from eacham_amd import ba
    L = oracle.lib()
    L.oracle_ba_solve.restype = C.c_int
    rc, out = ba.run_solver(lambda p, o, r, nt: L.oracle_ba_solve(p, o, r, nt), arrays, cfg, min_landmarks, trace_cap,
                            extra=(C.c_int(nthreads),), lm_factor=lm_factor)
    assert rc == 0, rc
    return out


# ---- triangulation ---------------------------------------------------------------------------

def tri_point(T1, T2, uv1, uv2, K4):
    L = oracle.lib()
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (np.reshape(T1, 16), np.reshape(T2, 16), uv1, uv2, K4)]
    out = np.zeros(3)
    L.oracle_triangulate_point.restype = None
    L.oracle_triangulate_point(*[C.c_void_p(x.ctypes.data) for x in a], C.c_void_p(out.ctypes.data))
    return out


def tri_angle(T1, T2, X):
    L = oracle.lib()
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (np.reshape(T1, 16), np.reshape(T2, 16), X)]
    L.oracle_triangulation_angle.restype = C.c_double
    return L.oracle_triangulation_angle(*[C.c_void_p(x.ctypes.data) for x in a])


def tri_tracks(transforms, track_ptr, obs_frame, obs_uv, K4, max_err, min_angle):
    L = oracle.lib()
    T = np.ascontiguousarray(transforms, dtype=np.float64).reshape(-1, 16)
    tp = np.ascontiguousarray(track_ptr, dtype=np.int32)
    of = np.ascontiguousarray(obs_frame, dtype=np.uint32)
    uv = np.ascontiguousarray(obs_uv, dtype=np.float64).reshape(-1, 2)
    K4 = np.ascontiguousarray(K4, dtype=np.float64)
    n = tp.size - 1
    pts = np.zeros((n, 3))
    status = np.zeros(n, dtype=np.int32)
    masks = np.zeros(of.size, dtype=np.uint8)
    vp = C.c_void_p
    rc = L.oracle_triangulate_tracks(vp(T.ctypes.data), C.c_int(n), vp(tp.ctypes.data), vp(of.ctypes.data), vp(uv.ctypes.data),
                                     vp(K4.ctypes.data), C.c_float(max_err), C.c_float(min_angle), vp(pts.ctypes.data),
                                     vp(status.ctypes.data), vp(masks.ctypes.data))
    if rc != 0:
        raise ValueError("oracle_triangulate_tracks: track_ptr is not monotone")
    return pts, status, masks


def reprojection_errors(transforms, frame, points, uv, K4):
    L = oracle.lib()
    T = np.ascontiguousarray(transforms, dtype=np.float64).reshape(-1, 16)
    fr = np.ascontiguousarray(frame, dtype=np.uint32)
    P = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    U = np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2)
    K4 = np.ascontiguousarray(K4, dtype=np.float64)
    err = np.zeros(fr.size, dtype=np.float32)
    vp = C.c_void_p
    L.oracle_reprojection_errors.restype = None
    L.oracle_reprojection_errors(vp(T.ctypes.data), C.c_int(fr.size), vp(fr.ctypes.data), vp(P.ctypes.data), vp(U.ctypes.data),
                                 vp(K4.ctypes.data), vp(err.ctypes.data))
    return err


# ---- view-graph query --------------------------------------------------------------------------

def graph_best_pair(n_frames, pairs, counts, offsets, q, t, valid, has3d_per_frame, excluded=None):
    from eacham_amd.graph import pack_has3d
    L = oracle.lib()
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    q = np.ascontiguousarray(q, dtype=np.uint32)
    t = np.ascontiguousarray(t, dtype=np.uint32)
    valid = np.ascontiguousarray(valid, dtype=np.uint8)
    excl = None if excluded is None else np.ascontiguousarray(excluded, dtype=np.uint8)
    kpo, flat = pack_has3d(has3d_per_frame)
    ec = np.zeros((pairs.shape[0], 2), dtype=np.uint32)
    best = np.zeros(3, dtype=np.uint32)
    vp = C.c_void_p
    L.oracle_graph_best_pair.restype = None
    L.oracle_graph_best_pair(C.c_int(n_frames), vp(pairs.ctypes.data), C.c_int(pairs.shape[0]), vp(counts.ctypes.data),
                             vp(offsets.ctypes.data), vp(q.ctypes.data), vp(t.ctypes.data), vp(valid.ctypes.data),
                             vp(excl.ctypes.data) if excl is not None else None, vp(kpo.ctypes.data), vp(flat.ctypes.data),
                             vp(ec.ctypes.data), vp(best.ctypes.data))
    return (int(best[0]), int(best[1]), int(best[2])), ec


def two_view_points(uv1, uv2, K4, transforms, max_err, min_angle, angle_strict):
    L = oracle.lib()
    U1 = np.ascontiguousarray(uv1, dtype=np.float64).reshape(-1, 2)
    U2 = np.ascontiguousarray(uv2, dtype=np.float64).reshape(-1, 2)
    T = np.ascontiguousarray(transforms, dtype=np.float64).reshape(-1, 16)
    K4 = np.ascontiguousarray(K4, dtype=np.float64)
    n, nt = U1.shape[0], T.shape[0]
    pts = np.zeros((nt, n, 3))
    keep = np.zeros((nt, n), dtype=np.uint8)
    counts = np.zeros(nt, dtype=np.int32)
    vp = C.c_void_p
    L.oracle_two_view_points.restype = None
    L.oracle_two_view_points(C.c_int(n), vp(U1.ctypes.data), vp(U2.ctypes.data), vp(K4.ctypes.data), C.c_int(nt), vp(T.ctypes.data),
                             C.c_float(max_err), C.c_float(min_angle), C.c_int(int(bool(angle_strict))), vp(pts.ctypes.data),
                             vp(keep.ctypes.data), vp(counts.ctypes.data))
    return pts, keep, counts


def score_hypotheses(kind, a, b, models, K=None, threshold=16.0):
    """oracle/score_oracle.c: (errors [nm, n] float32, inlier counts, medians)."""
    from eacham_amd import score
    L = oracle.lib()
    k, a, b, models, K4 = score.marshal(kind, a, b, models, K)
    n, nm = a.shape[0], models.shape[0]
    err = np.zeros((nm, n), dtype=np.float32)
    counts = np.zeros(nm, dtype=np.int32)
    med = np.zeros(nm, dtype=np.float32)
    vp = C.c_void_p
    L.oracle_score_hypotheses.restype = None
    L.oracle_score_hypotheses(C.c_int(k), C.c_int(n), vp(a.ctypes.data), vp(b.ctypes.data), C.c_int(nm), vp(models.ctypes.data),
                              vp(K4.ctypes.data) if K4 is not None else None, C.c_float(threshold), vp(err.ctypes.data),
                              vp(counts.ctypes.data), vp(med.ctypes.data))
    return err, counts, med


def solve_minimal(kind, a, b, samples, K=None):
    """oracle_solve_minimal (oracle/solve_oracle.c): kind "homography4" / "essential5"; same contract as eacham_solve_minimal."""
    L = oracle.lib()
    k, m, maxm = {"homography4": (0, 4, 1), "essential5": (1, 5, 10)}[kind]
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 2)
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1, 2)
    idx = np.ascontiguousarray(samples, dtype=np.int32).reshape(-1, m)
    K4 = None if K is None else np.ascontiguousarray(K, dtype=np.float64).reshape(4)
    models = np.zeros((idx.shape[0], maxm, 9), dtype=np.float64)
    counts = np.zeros(idx.shape[0], dtype=np.int32)
    vp = C.c_void_p
    L.oracle_solve_minimal.restype = None
    L.oracle_solve_minimal(C.c_int(k), vp(a.ctypes.data), vp(b.ctypes.data), vp(K4.ctypes.data) if K4 is not None else None,
                           C.c_int(idx.shape[0]), vp(idx.ctypes.data), vp(models.ctypes.data), vp(counts.ctypes.data))
    return models, counts


def solve_pnp(object_points, image_points, K, samples):
    """oracle_solve_pnp (oracle/solve_oracle.c): EPnP per row of `samples`; same contract as eacham_solve_pnp."""
    L = oracle.lib()
    X = np.ascontiguousarray(object_points, dtype=np.float64).reshape(-1, 3)
    uv = np.ascontiguousarray(image_points, dtype=np.float64).reshape(-1, 2)
    idx = np.ascontiguousarray(samples, dtype=np.int32)
    K4 = np.ascontiguousarray(K, dtype=np.float64).reshape(4)
    models = np.zeros((idx.shape[0], 12), dtype=np.float64)
    ok = np.zeros(idx.shape[0], dtype=np.int32)
    vp = C.c_void_p
    L.oracle_solve_pnp.restype = None
    L.oracle_solve_pnp(vp(X.ctypes.data), vp(uv.ctypes.data), vp(K4.ctypes.data), C.c_int(idx.shape[1]), C.c_int(idx.shape[0]),
                       vp(idx.ctypes.data), vp(models.ctypes.data), vp(ok.ctypes.data))
    return models, ok
