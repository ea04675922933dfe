"""Seeded two-view / PnP hypothesis sets for the scoring tests (shared by the CPU and the GPU test files)."""
import numpy as np

from eacham_amd import synth


def skew(t):
    return np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])


def two_view_case(n=700, n_models=40, seed=7, outliers=0.25, planar=False, facing=False, noise=0.5):
    """Two cameras of a synth scene seeing n landmarks: pixel correspondences (float32-valued, as cv::Point2f),
    the true E / H and n_models candidates around them (the first is the truth, the rest perturbed, some garbage)."""
    sc = synth.make_scene(2, n, 2, seed=seed, pixel_noise=0.7)
    K = sc["K"]
    T1, T2 = sc["T_true"][0], sc["T_true"][1]
    X = sc["points_true"].copy()
    if planar:
        if facing:                                          # a plane both cameras (at x = 4) see from the same side
            X[:, 0] = 0.05 * X[:, 1] - 0.03 * X[:, 2]
        else:                                               # points on a plane: a homography maps view 1 to view 2
            X[:, 2] = 0.05 * X[:, 0] - 0.03 * X[:, 1]       # (the cameras sit on opposite sides of this one: fine for scoring only)
    rng = np.random.default_rng(seed)

    def proj(T):
        pc = X @ T[:3, :3].T + T[:3, 3]
        return np.stack([K[0] * pc[:, 0] / pc[:, 2] + K[2], K[1] * pc[:, 1] / pc[:, 2] + K[3]], 1)
    uv1 = proj(T1) + noise * rng.normal(size=(n, 2))
    uv2 = proj(T2) + noise * rng.normal(size=(n, 2))
    bad = rng.random(n) < outliers
    uv2[bad] += rng.normal(0, 60, size=(int(bad.sum()), 2))
    uv1, uv2 = uv1.astype(np.float32).astype(np.float64), uv2.astype(np.float32).astype(np.float64)
    T21 = T2 @ np.linalg.inv(T1)
    R, t = T21[:3, :3], T21[:3, 3]
    E = skew(t / np.linalg.norm(t)) @ R
    Kmat = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1.0]])
    if planar:
        # plane n.X = d in camera 1: H = K (R + t n^T / d) K^-1, from three plane points
        Xc = X[:3] @ T1[:3, :3].T + T1[:3, 3]
        nrm = np.cross(Xc[1] - Xc[0], Xc[2] - Xc[0])
        d = nrm @ Xc[0]
        H = Kmat @ (R + np.outer(t, nrm) / d) @ np.linalg.inv(Kmat)
        H = H / H[2, 2]
    else:
        H = np.eye(3)
    Es, Hs = [E], [H]
    for m in range(1, n_models):
        s = 10.0 ** rng.uniform(-4, -0.5)
        Es.append(E + s * rng.normal(size=(3, 3)) if m % 7 else rng.normal(size=(3, 3)))
        Hm = H + s * rng.normal(size=(3, 3)) * np.array([[1, 1, 50], [1, 1, 50], [1e-3, 1e-3, 0]])
        Hs.append(Hm / Hm[2, 2])
    return {"K": K, "uv1": uv1, "uv2": uv2, "E": np.array(Es).reshape(-1, 9), "H": np.array(Hs).reshape(-1, 9), "bad": bad, "T21": T21}


def pnp_case(n=500, n_models=64, seed=11, outliers=0.3):
    """Object points + their pixels in one camera; candidates = the true [R|t] and perturbed / random poses."""
    sc = synth.make_scene(3, n, 3, seed=seed)
    K, T = sc["K"], sc["T_true"][1]
    X = sc["points_true"]
    rng = np.random.default_rng(seed)
    pc = X @ T[:3, :3].T + T[:3, 3]
    uv = np.stack([K[0] * pc[:, 0] / pc[:, 2] + K[2], K[1] * pc[:, 1] / pc[:, 2] + K[3]], 1) + 0.8 * rng.normal(size=(n, 2))
    bad = rng.random(n) < outliers
    uv[bad] += rng.normal(0, 80, size=(int(bad.sum()), 2))
    uv = uv.astype(np.float32).astype(np.float64)
    models = [np.concatenate([T[:3, :3].reshape(-1), T[:3, 3]])]
    for m in range(1, n_models):
        w = 10.0 ** rng.uniform(-4, -0.3) * rng.normal(size=3)
        R = synth.so3_exp(w) @ T[:3, :3]
        t = T[:3, 3] + 10.0 ** rng.uniform(-4, -0.5) * rng.normal(size=3)
        if m % 9 == 0:
            t = -t                                  # points behind the camera: huge / negative-depth projections
        models.append(np.concatenate([R.reshape(-1), t]))
    models = np.array(models)
    X = X.copy()
    X[0] = -T[:3, :3].T @ T[:3, 3]                  # an object point AT the camera centre of the true pose: z = 0 -> 1/z := 1
    return {"K": K, "X": X, "uv": uv, "models": models, "bad": bad}


def planar_pnp_case(n=300, seed=3, noise=0.0):
    """COPLANAR object points (a tilted plane through the scene of pnp_case), their pixels under a known pose + `noise` px of
    Gaussian noise, K, that pose (R row-major | t). The planar form of EPnP (oracle/solve_oracle.c's header) is what solves these."""
    c = pnp_case(n=n, seed=seed, outliers=0.0)
    X, T, K = np.ascontiguousarray(c["X"][1:]).copy(), c["models"][0], np.asarray(c["K"], float)
    X[:, 2] = 0.3 * X[:, 0] - 0.2 * X[:, 1] + 1.0
    pc = X @ T[:9].reshape(3, 3).T + T[9:]
    assert pc[:, 2].min() > 0.5
    uv = np.stack([K[0] * pc[:, 0] / pc[:, 2] + K[2], K[1] * pc[:, 1] / pc[:, 2] + K[3]], 1)
    rng = np.random.default_rng(seed + 1000)
    return X, uv + noise * rng.normal(size=uv.shape), K, T
