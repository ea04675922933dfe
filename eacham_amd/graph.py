"""Host-side mirror of the view-graph query on the CSR match graph.

  best_pair_for_valid(...)  <->  Graph::GetBestPairForValid (modules/sfm/data/Graph.h:59-106)

The match graph is the wire format `HipContext.match_all_pairs` returns: (pairs, counts, offsets, q, t);
pair p with counts[p] > 0 is the factor f1 -> f2 (matches q -> t) and the factor f2 -> f1 (t -> q),
i.e. the two Graph::Connect calls of apps/sfm/main.cpp:144-145.
"""
from __future__ import annotations

import numpy as np

from .matcher import HipContext

NONE = 0xFFFFFFFF  # std::numeric_limits<unsigned>::max() of the reference's empty result


def pack_has3d(per_frame) -> tuple[np.ndarray, np.ndarray]:
    """per_frame[f] = bool array over the keypoints of frame f (HasPoint3d && !IsPoint3dTwoView)."""
    sizes = np.array([len(a) for a in per_frame], dtype=np.int64)
    kp_offsets = np.zeros(len(per_frame) + 1, dtype=np.int64)
    kp_offsets[1:] = np.cumsum(sizes)
    flat = np.concatenate([np.asarray(a, dtype=np.uint8) for a in per_frame]) if len(per_frame) else np.zeros(0, np.uint8)
    return kp_offsets, np.ascontiguousarray(flat, dtype=np.uint8)


def best_pair_for_valid(ctx: HipContext, n_frames: int, pairs, counts, offsets, q, t, valid, has3d_per_frame,
                        excluded=None, want_edge_counts: bool = False):
    """Returns (id, id2, points3dCount) [, edge_counts npairs x 2]."""
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    q = np.ascontiguousarray(q, dtype=np.uint32)
    t = np.ascontiguousarray(t, dtype=np.uint32)
    valid = np.ascontiguousarray(valid, dtype=np.uint8)
    excl = None if excluded is None else np.ascontiguousarray(excluded, dtype=np.uint8)
    kp_offsets, flat = pack_has3d(has3d_per_frame)
    if valid.size != n_frames or kp_offsets.size != n_frames + 1 or (excl is not None and excl.size != n_frames):
        raise ValueError("per-frame arrays must have n_frames entries")
    ec = np.zeros((pairs.shape[0], 2), dtype=np.uint32)
    best = np.zeros(3, dtype=np.uint32)
    ctx._check(ctx._L.eacham_graph_best_pair(
        ctx.handle, n_frames, pairs.ctypes.data, pairs.shape[0], counts.ctypes.data, offsets.ctypes.data, q.ctypes.data,
        t.ctypes.data, valid.ctypes.data, excl.ctypes.data if excl is not None else None, kp_offsets.ctypes.data,
        flat.ctypes.data, ec.ctypes.data, best.ctypes.data))
    out = (int(best[0]), int(best[1]), int(best[2]))
    return (out, ec) if want_edge_counts else out
