"""Deterministic synthetic SfM scenes and descriptors.

The reference ships no datasets and none are reachable offline, so every benchmark / parity input
is generated here from a counter-based RNG (splitmix64) that depends only on (seed, stream, index):
the same arrays come out on every rank and every box, independent of numpy's generator versions.

Scene layout mirrors what eacham's app would hold after feature extraction and triangulation:
  * cameras  — world->camera 4x4 transforms (`Node::transform`, modules/sfm/data/Node.h:215-228)
  * K        — fx = fy = 1.2*max(w,h), cx = w/2, cy = h/2 (modules/sfm/utils/Utils.h:13-22)
  * map      — landmark positions + observer lists (modules/sfm/data/Map.h:15-23)
  * frames   — N x D fp32 descriptor matrices, SIFT-like integer values in [0,255]
               (modules/base/features/FeatureExtractorSift.cpp:14-26)
"""
from __future__ import annotations

import numpy as np

MASTER_SEED = 12345  # homage to modules/sfm/reconstruction/Triangulator.cpp:192

_U64 = np.uint64
_MASK53 = (1 << 53) - 1


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + _U64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        z = z ^ (z >> _U64(31))
    return z


def rng_u64(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """Counter-based 64-bit words: word k of sub-stream `stream` of `seed`."""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = _splitmix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ _splitmix64(np.uint64(stream)))
        return _splitmix64(key + idx * _U64(0xD1342543DE82EF95))


def rng_uniform(seed: int, stream: int, shape) -> np.ndarray:
    """U[0,1) doubles with 53 random bits."""
    n = int(np.prod(shape))
    w = rng_u64(seed, stream, np.arange(n, dtype=np.uint64))
    return ((w >> _U64(11)).astype(np.float64) * (1.0 / (1 << 53))).reshape(shape)


def rng_normal(seed: int, stream: int, shape) -> np.ndarray:
    """N(0,1) doubles (Box-Muller on two sub-streams)."""
    u1 = rng_uniform(seed, 2 * stream + 1_000_003, shape)
    u2 = rng_uniform(seed, 2 * stream + 1_000_004, shape)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return r * np.cos(2.0 * np.pi * u2)


def rng_permutation(seed: int, stream: int, n: int) -> np.ndarray:
    w = rng_u64(seed, stream, np.arange(n, dtype=np.uint64))
    return np.argsort(w, kind="stable")


# --------------------------------------------------------------------------------------------
# geometry helpers
# --------------------------------------------------------------------------------------------

def look_at_world_to_cam(center: np.ndarray, target: np.ndarray) -> np.ndarray:
    """World->camera 4x4 with +z looking from `center` to `target`, image y pointing down."""
    z = target - center
    z = z / np.linalg.norm(z)
    up = np.array([0.0, 0.0, 1.0])
    x = np.cross(z, up)
    if np.linalg.norm(x) < 1e-9:
        x = np.array([1.0, 0.0, 0.0])
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    R_wc = np.stack([x, y, z], axis=0)  # rows = camera axes in world coords
    T = np.eye(4)
    T[:3, :3] = R_wc
    T[:3, 3] = -R_wc @ center
    return T


def so3_exp(w: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * (K @ K)


def make_scene(n_cams: int = 200, n_landmarks: int = 50_000, k_obs: int = 10,
               seed: int = MASTER_SEED, image_size: int = 800, pixel_noise: float = 1.0,
               rot_noise: float = 0.01, trans_noise: float = 0.02, point_noise: float = 0.02):
    """The metric scene S200 (defaults) or any scaled variant.

    Cameras on a 2-turn helix of radius 4 and height -1..+1 looking at the origin; landmarks uniform
    in [-1,1]^3; each landmark is observed by its `k_obs` nearest cameras (positive depth is
    guaranteed by the geometry); uv = projection + N(0, pixel_noise). The BA start perturbs poses
    by N(0, rot_noise) rad / N(0, trans_noise) and points by N(0, point_noise); K stays exact;
    camera 0 is the fixed node (`Graph::FixNode`, modules/sfm/utils/Utils.h:38).
    """
    f = 1.2 * image_size
    K = np.array([f, f, image_size / 2.0, image_size / 2.0])  # fx, fy, cx, cy
    s = np.arange(n_cams) / max(n_cams - 1, 1)
    ang = 4.0 * np.pi * s
    centers = np.stack([4.0 * np.cos(ang), 4.0 * np.sin(ang), -1.0 + 2.0 * s], axis=1)
    T_true = np.stack([look_at_world_to_cam(c, np.zeros(3)) for c in centers])
    pts_true = rng_uniform(seed, 1, (n_landmarks, 3)) * 2.0 - 1.0

    k_obs = min(k_obs, n_cams)
    d2 = ((pts_true[:, None, :] - centers[None, :, :]) ** 2).sum(-1)  # Nl x Nc
    cam_sel = np.argsort(d2, axis=1, kind="stable")[:, :k_obs]
    cam_sel.sort(axis=1)
    lm_idx = np.repeat(np.arange(n_landmarks), k_obs)
    cam_idx = cam_sel.reshape(-1)
    R = T_true[cam_idx, :3, :3]
    t = T_true[cam_idx, :3, 3]
    pc = np.einsum("nij,nj->ni", R, pts_true[lm_idx]) + t
    assert (pc[:, 2] > 0).all()
    uv = np.stack([K[0] * pc[:, 0] / pc[:, 2] + K[2], K[1] * pc[:, 1] / pc[:, 2] + K[3]], axis=1)
    uv = uv + pixel_noise * rng_normal(seed, 2, uv.shape)

    # BA initial guess
    T_init = T_true.copy()
    dw = rot_noise * rng_normal(seed, 3, (n_cams, 3))
    dt = trans_noise * rng_normal(seed, 4, (n_cams, 3))
    for i in range(1, n_cams):  # camera 0 is fixed and exact
        T_init[i, :3, :3] = so3_exp(dw[i]) @ T_true[i, :3, :3]
        T_init[i, :3, 3] = T_true[i, :3, 3] + dt[i]
    pts_init = pts_true + point_noise * rng_normal(seed, 5, pts_true.shape)
    fixed = np.zeros(n_cams, dtype=np.int32)
    fixed[0] = 1
    return {
        "K": K,
        "T_true": T_true,
        "T_init": T_init,
        "points_true": pts_true,
        "points_init": pts_init,
        "fixed": fixed,
        "obs_cam": cam_idx.astype(np.uint32),
        "obs_lm": lm_idx.astype(np.uint32),
        "obs_uv": uv,
        "observers": np.full(n_landmarks, k_obs, dtype=np.int32),
    }


def local_window(scene, frame: int, min_shared: int = 31, max_neighbours: int | None = None):
    """The local BA window RefineBA(currentFrameId >= 0) builds (BundleAdjuster.cpp:123-145): the current
    frame first, then its valid neighbours in the view graph in ascending id — here the frames that share at
    least `min_shared` landmarks with it (an edge of the match graph needs more than 30 mutual matches,
    apps/sfm/main.cpp:142). Landmarks = every map point a window frame observes (first-seen order, :100-117);
    `observers` keeps the GLOBAL observer counts (:109). Returns a scene-shaped dict (BaArrays.from_scene works)
    plus "frames" (window index -> frame id) and "landmarks" (window index -> landmark id)."""
    obs_cam, obs_lm = scene["obs_cam"].astype(np.int64), scene["obs_lm"].astype(np.int64)
    n_cams = scene["T_true"].shape[0]
    mine = obs_lm[obs_cam == frame]
    seen = np.zeros(scene["points_true"].shape[0], dtype=bool)
    seen[mine] = True
    shared = np.bincount(obs_cam[seen[obs_lm]], minlength=n_cams)
    nb = [int(c) for c in np.nonzero(shared >= min_shared)[0] if c != frame]
    if max_neighbours is not None:
        nb = sorted(sorted(nb, key=lambda c: -shared[c])[:max_neighbours])
    frames = np.array([frame] + nb, dtype=np.int64)
    cam_new = np.full(n_cams, -1, dtype=np.int64)
    cam_new[frames] = np.arange(frames.size)
    sel = np.nonzero(cam_new[obs_cam] >= 0)[0]
    # frameAdder visits the window frames in order, each frame's points in ascending keypoint (= observation) order
    sel = sel[np.lexsort((sel, cam_new[obs_cam[sel]]))]
    lm_ids, first = np.unique(obs_lm[sel], return_index=True)
    lm_ids = lm_ids[np.argsort(first, kind="stable")]  # first-seen order
    lm_new = np.full(scene["points_true"].shape[0], -1, dtype=np.int64)
    lm_new[lm_ids] = np.arange(lm_ids.size)
    return {
        "K": scene["K"].copy(),
        "T_true": scene["T_true"][frames], "T_init": scene["T_init"][frames],
        "points_true": scene["points_true"][lm_ids], "points_init": scene["points_init"][lm_ids],
        "fixed": scene["fixed"][frames].copy(),
        "obs_cam": cam_new[obs_cam[sel]].astype(np.uint32), "obs_lm": lm_new[obs_lm[sel]].astype(np.uint32),
        "obs_uv": scene["obs_uv"][sel], "observers": scene["observers"][lm_ids].copy(),
        "frames": frames, "landmarks": lm_ids,
    }


# --------------------------------------------------------------------------------------------
# descriptors
# --------------------------------------------------------------------------------------------

def _landmark_bases(lm_ids: np.ndarray, dim: int, seed: int) -> np.ndarray:
    """SIFT-like base vector per landmark id: round(clip(|N(0,48)|, 0, 255))."""
    idx = lm_ids.astype(np.uint64)[:, None] * _U64(dim) + np.arange(dim, dtype=np.uint64)[None, :]
    u1 = (rng_u64(seed, 11, idx) >> _U64(11)).astype(np.float64) * (1.0 / (1 << 53))
    u2 = (rng_u64(seed, 12, idx) >> _U64(11)).astype(np.float64) * (1.0 / (1 << 53))
    g = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return np.clip(np.rint(np.abs(48.0 * g)), 0, 255)


def make_frame_descriptors(scene, n_kpts: int = 2000, dim: int = 256, seed: int = MASTER_SEED,
                           frames=None, obs_noise: float = 8.0):
    """Per-frame N x dim fp32 descriptor matrices in mode U8 (integer values in [0,255]).

    Rows = the frame's observed landmarks (up to n_kpts) + distractors drawn from fresh landmark
    ids, shuffled. Returns (list of arrays, list of per-row landmark ids; -1 = distractor).
    """
    n_cams = scene["T_true"].shape[0]
    n_lm = scene["points_true"].shape[0]
    frames = range(n_cams) if frames is None else frames
    order = np.argsort(scene["obs_cam"], kind="stable")
    cam_sorted = scene["obs_cam"][order]
    starts = np.searchsorted(cam_sorted, np.arange(n_cams + 1))
    descs, ids = [], []
    for f in frames:
        lm = scene["obs_lm"][order[starts[f]:starts[f + 1]]].astype(np.int64)
        if lm.size > n_kpts:
            lm = lm[rng_permutation(seed, 100_000 + f, lm.size)[:n_kpts]]
        n_dis = n_kpts - lm.size
        dis = n_lm + f * n_kpts + np.arange(n_dis, dtype=np.int64)  # ids never shared
        all_ids = np.concatenate([lm, dis])
        base = _landmark_bases(all_ids, dim, seed)
        noise = np.rint(obs_noise * rng_normal(seed, 200_000 + f, base.shape))
        d = np.clip(base + noise, 0, 255)
        perm = rng_permutation(seed, 300_000 + f, n_kpts)
        descs.append(np.ascontiguousarray(d[perm], dtype=np.float32))
        tag = np.concatenate([lm, np.full(n_dis, -1, dtype=np.int64)])
        ids.append(tag[perm])
    return descs, ids


def random_u8_descriptors(n: int, dim: int, seed: int, stream: int = 0, spread: float = 48.0) -> np.ndarray:
    """Unstructured integer-valued descriptors (for unit tests)."""
    g = rng_normal(seed, 400_000 + stream, (n, dim))
    return np.ascontiguousarray(np.clip(np.rint(np.abs(spread * g)), 0, 255), dtype=np.float32)


def unit_float_descriptors(n: int, dim: int, seed: int, stream: int = 0, shared: np.ndarray | None = None,
                           noise: float = 0.15) -> np.ndarray:
    """SuperPoint / LightGlue style descriptors (mode F32 of SURVEY.md §8(d)): unit-norm fp32 rows.
    `shared` (m x dim, unit rows) seeds the first m rows with noisy copies, i.e. true correspondences."""
    g = rng_normal(seed, 500_000 + stream, (n, dim))
    if shared is not None:
        m = min(n, shared.shape[0])
        g[:m] = shared[:m] * np.sqrt(dim) + noise * rng_normal(seed, 600_000 + stream, (m, dim))
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    return np.ascontiguousarray(g, dtype=np.float32)


def all_pairs(n_frames: int) -> np.ndarray:
    """Unordered frame pairs (i<j), the unit of the pair loop in apps/sfm/main.cpp:84-92."""
    i, j = np.triu_indices(n_frames, k=1)
    return np.ascontiguousarray(np.stack([i, j], axis=1), dtype=np.int32)


# --------------------------------------------------------------------------------------------
# triangulation tracks
# --------------------------------------------------------------------------------------------

def make_tracks(scene, n_tracks: int | None = None, seed: int = MASTER_SEED, min_obs: int = 2, max_obs: int | None = None,
                outlier_frac: float = 0.15, outlier_px: float = 30.0, use_true_poses: bool = True):
    """Candidate tracks as TriangulateFrame builds them (Triangulator.cpp:248-262): per landmark a
    subset of its observers (sorted by frame id, as the reference's std::map iterates), the noisy
    pixels of the scene, and for `outlier_frac` of the tracks one observation displaced by
    `outlier_px`. Returns dict(transforms n_cams x 16, track_ptr, obs_frame, obs_uv, K, landmark)."""
    n_lm = scene["points_true"].shape[0]
    k = int(scene["observers"][0])
    max_obs = k if max_obs is None else min(max_obs, k)
    n_tracks = n_lm if n_tracks is None else min(n_tracks, n_lm)
    cam = scene["obs_cam"].reshape(n_lm, k)[:n_tracks]
    uv = scene["obs_uv"].reshape(n_lm, k, 2)[:n_tracks]
    m = min_obs + (rng_u64(seed, 700_001, np.arange(n_tracks)) % _U64(max_obs - min_obs + 1)).astype(np.int64)
    rank = np.argsort(rng_u64(seed, 700_002, np.arange(n_tracks * k)).reshape(n_tracks, k), axis=1, kind="stable")
    keep = rank < m[:, None]  # m random observers per track, original (frame-sorted) order kept
    track_ptr = np.zeros(n_tracks + 1, dtype=np.int32)
    track_ptr[1:] = np.cumsum(m)
    obs_frame = cam[keep].astype(np.uint32)
    obs_uv = uv[keep].copy()
    bad = rng_uniform(seed, 700_003, (n_tracks,)) < outlier_frac
    which = (rng_u64(seed, 700_004, np.arange(n_tracks)) % m.astype(np.uint64)).astype(np.int64)
    sel = track_ptr[:-1][bad] + which[bad]
    obs_uv[sel] += outlier_px
    T = scene["T_true"] if use_true_poses else scene["T_init"]
    return {"transforms": np.ascontiguousarray(T.reshape(-1, 16)), "track_ptr": track_ptr, "obs_frame": obs_frame,
            "obs_uv": np.ascontiguousarray(obs_uv), "K": scene["K"].copy(), "landmark": np.arange(n_tracks)}
