"""Independent numpy statement of the matching semantics (a third opinion beside oracle/ and HIP)."""
import numpy as np


def sq_dists(A, B):
    a = np.rint(np.asarray(A)).astype(np.int64)
    b = np.rint(np.asarray(B)).astype(np.int64)
    return (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2 * (a @ b.T)


def directed(A, B, ratio=0.8):
    """q -> t for rows passing Lowe's test; exact 2-NN, ties -> lower train index."""
    n1, n2 = len(A), len(B)
    if n1 == 0 or n2 < 2:
        return np.zeros(0, np.uint32), np.zeros(0, np.uint32)
    D = sq_dists(A, B)
    order = np.argsort(D, axis=1, kind="stable")[:, :2]
    d0 = np.sqrt(np.take_along_axis(D, order[:, :1], 1)[:, 0].astype(np.float32))
    d1 = np.sqrt(np.take_along_axis(D, order[:, 1:2], 1)[:, 0].astype(np.float32))
    with np.errstate(divide="ignore", invalid="ignore"):
        quot = (d0 / d1).astype(np.float32)
    ok = quot.astype(np.float64) < ratio
    q = np.nonzero(ok)[0].astype(np.uint32)
    return q, order[ok, 0].astype(np.uint32)


def mutual(A, B, ratio=0.8, min_dir=30, min_mutual=30):
    q12, t12 = directed(A, B, ratio)
    q21, t21 = directed(B, A, ratio)
    back = dict(zip(q21.tolist(), t21.tolist()))
    keep = [(q, t) for q, t in zip(q12.tolist(), t12.tolist()) if back.get(t, -1) == q]
    edge = len(q12) >= min_dir and len(q21) >= min_dir and len(keep) > min_mutual
    stats = np.array([len(q12), len(q21), len(keep), int(edge)], dtype=np.int32)
    if not edge:
        keep = []
    q = np.array([k[0] for k in keep], dtype=np.uint32)
    t = np.array([k[1] for k in keep], dtype=np.uint32)
    return q, t, stats


# ---- bundle adjustment: independent numpy statement of graph.error --------
def _huber(n, k):
    return np.where(n <= k, 0.5 * n * n, k * (n - 0.5 * k))


def _cayley_local(R):
    return 2.0 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (1.0 + np.trace(R))


def ba_error_np(arrays, T_wc=None, points=None, K=None):
    """Sum of factor errors of RefineBA's graph at the given values (priors centred on `arrays`)."""
    T0 = np.asarray(arrays.cam_T_wc, np.float64).reshape(-1, 4, 4)
    P0 = np.asarray(arrays.points, np.float64).reshape(-1, 3)
    K0 = np.array([arrays.K[0], arrays.K[1], 0.0, arrays.K[2], arrays.K[3]])
    T = T0 if T_wc is None else np.asarray(T_wc, np.float64).reshape(-1, 4, 4)
    P = P0 if points is None else np.asarray(points, np.float64).reshape(-1, 3)
    Kv = K0 if K is None else np.array([K[0], K[1], 0.0, K[2], K[3]])
    f32 = np.float32
    err = 0.0
    # reprojection factors: sigma 1.5 px, Huber 3.0 on the whitened 2-vector norm
    c, j = arrays.obs_cam.astype(int), arrays.obs_point.astype(int)
    pc = np.einsum("nij,nj->ni", T[c, :3, :3], P[j]) + T[c, :3, 3]
    ok = pc[:, 2] > 0
    u, v = pc[:, 0] / pc[:, 2], pc[:, 1] / pc[:, 2]
    r = np.stack([Kv[0] * u + Kv[2] * v + Kv[3], Kv[1] * v + Kv[4]], 1) - arrays.obs_uv
    n = np.where(ok, np.linalg.norm(r, axis=1) / 1.5, 0.0)
    err += _huber(n, 3.0).sum()
    # pose priors
    rot_sig = float(f32(45.0) * f32(3.141592) / f32(180.0))
    fix_rot = float(f32(0.0001) * f32(3.141592) / f32(180.0))
    for i in range(len(T)):
        Rx, tx = T[i, :3, :3].T, -T[i, :3, :3].T @ T[i, :3, 3]      # camera->world
        Rp, tp = T0[i, :3, :3].T, -T0[i, :3, :3].T @ T0[i, :3, 3]
        xi = np.concatenate([_cayley_local(Rx.T @ Rp), Rx.T @ (tp - tx)])
        if arrays.cam_fixed[i]:
            sg = np.array([fix_rot] * 3 + [float(f32(0.0001))] * 3)
            err += 0.5 * np.sum((xi / sg) ** 2)
        else:
            sg = np.array([rot_sig] * 3 + [float(f32(0.35))] * 3)
            err += float(_huber(np.linalg.norm(xi / sg), 2.5))
    # landmark priors (only landmarks that appear in the graph)
    used = np.zeros(len(P), bool)
    used[j] = True
    obs = np.maximum(arrays.point_observers.astype(np.float32), 1)
    sg = (f32(1.0) / obs).astype(np.float64)
    kh = (f32(3.0) / obs).astype(np.float64)
    n = np.linalg.norm(P - P0, axis=1) / sg
    err += _huber(n, kh)[used].sum()
    # calibration prior
    err += 0.5 * np.sum(((Kv - K0) / np.array([25, 25, 1e-5, 1e-4, 1e-4])) ** 2)
    return float(err)


# ---- triangulation (numpy SVD as the third opinion) ------------------------------------------

def tri_point(T1, T2, uv1, uv2, K4):
    """Triangulator.cpp:49-77 with numpy's LAPACK SVD instead of a Jacobi SVD."""
    T1, T2 = np.reshape(T1, (4, 4)), np.reshape(T2, (4, 4))
    x1, y1 = (uv1[0] - K4[2]) / K4[0], (uv1[1] - K4[3]) / K4[1]
    x2, y2 = (uv2[0] - K4[2]) / K4[0], (uv2[1] - K4[3]) / K4[1]
    A = np.stack([y1 * T1[2] - T1[1], x1 * T1[2] - T1[0], y2 * T2[2] - T2[1], x2 * T2[2] - T2[0]])
    v = np.linalg.svd(A)[2][3]
    return v[:3] / v[3]


def tri_angle(T1, T2, X):
    c1 = np.linalg.inv(np.reshape(T1, (4, 4)))[:3, 3]
    c2 = np.linalg.inv(np.reshape(T2, (4, 4)))[:3, 3]
    r1, r2 = X - c1, X - c2
    n1, n2 = np.linalg.norm(r1), np.linalg.norm(r2)
    if n1 < np.float32(0.0000001) or n2 < np.float32(0.0000001):
        return 0.0
    a = np.arccos(r1 @ r2 / (n1 * n2))
    return min(a, np.pi - a)


def tri_inlier(T, uv, K4, X, max_err):
    T = np.reshape(T, (4, 4))
    p = T[:3, :3] @ X + T[:3, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        u, v = K4[0] * p[0] / p[2] + K4[2], K4[1] * p[1] / p[2] + K4[3]
        err = np.float32(np.sqrt((uv[0] - u) ** 2 + (uv[1] - v) ** 2))
    return bool(err < np.float32(max_err)) and bool(T[2] @ np.append(X, 1.0) >= np.finfo(float).eps)


def tri_ransac(Ts, uvs, K4, max_err, min_angle):
    """Triangulator.cpp:96-186 literally. Returns (ok, point, mask list)."""
    m = len(Ts)
    if m < 2:
        return False, np.zeros(3), []
    if m < 3:
        X = tri_point(Ts[0], Ts[1], uvs[0], uvs[1], K4)
        if tri_angle(Ts[0], Ts[1], X) < np.float32(min_angle):
            return False, X, []
        mask = [tri_inlier(Ts[i], uvs[i], K4, X, max_err) for i in range(m)]
        return bool(X[2] > 0), X, mask
    best, mask, X = 0, [], np.zeros(3)
    for r1 in range(m - 1):
        for r2 in range(r1 + 1, m):
            X = tri_point(Ts[r1], Ts[r2], uvs[r1], uvs[r2], K4)
            if tri_angle(Ts[r1], Ts[r2], X) >= np.float32(min_angle):
                loc = [tri_inlier(Ts[i], uvs[i], K4, X, max_err) for i in range(m)]
                if sum(loc) > best:
                    best, mask = sum(loc), loc
    return bool(X[2] > 0 and best > 2), X, mask


# ---- view-graph query (Graph.h:59-106 walked literally on dict-of-dict factors) ----------------

def graph_best_pair(n_frames, pairs, counts, offsets, q, t, valid, has3d, excluded=None):
    factors = {f: {} for f in range(n_frames)}
    for p, (f1, f2) in enumerate(np.asarray(pairs).reshape(-1, 2).tolist()):
        if counts[p] > 0:
            sl = slice(int(offsets[p]), int(offsets[p]) + int(counts[p]))
            factors[f1][f2] = np.asarray(q[sl])          # m1 of the factor f1 -> f2
            factors[f2][f1] = np.asarray(t[sl])          # m1 of the factor f2 -> f1
    best_score, best = 0.0, (0xFFFFFFFF, 0xFFFFFFFF, 0)
    for node in range(n_frames):                        # std::map: ascending id
        if not valid[node]:
            continue
        for other in sorted(factors[node]):             # neighbours in ascending id (documented choice)
            if valid[other] or (excluded is not None and excluded[other]):
                continue
            cnt = int(np.asarray(has3d[node])[factors[node][other]].sum())
            if best_score > cnt:
                continue
            best_score, best = float(cnt), (node, other, cnt)
    return best


def two_view(uv1, uv2, K4, T, max_err, min_angle, strict):
    """ReconstructionManager.cpp:118-143 (strict) / :162-186 per match, with LAPACK's SVD."""
    I4 = np.eye(4)
    T = np.reshape(T, (4, 4))
    pts, keep = [], []
    for p1, p2 in zip(uv1, uv2):
        X = tri_point(I4, T, p1, p2, K4)
        ok = False
        if not (X[2] <= 0.0):
            u, v = K4[0] * X[0] / X[2] + K4[2], K4[1] * X[1] / X[2] + K4[3]
            err = np.float32(np.sqrt((p1[0] - u) ** 2 + (p1[1] - v) ** 2))
            ang = tri_angle(I4, T, X)
            ok = bool(err < np.float32(max_err)) and (ang > np.float32(min_angle) if strict else not (ang < np.float32(min_angle)))
        pts.append(X)
        keep.append(ok)
    return np.array(pts), np.array(keep)
