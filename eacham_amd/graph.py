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


class ResidentGraph:
    """eacham_graph_create / _set_frame / _query: the match graph uploaded once, the per-frame state set frame by frame,
    the query two small kernels — what the incremental loop of apps/sfm/main.cpp:188-214 uses after every frame it adds."""

    def __init__(self, ctx: HipContext, n_frames: int, pairs, counts, offsets, q, t, keypoints_per_frame):
        import ctypes as C
        self._C, self.ctx, self.n_frames = C, ctx, n_frames
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        q = np.ascontiguousarray(q, dtype=np.uint32)
        t = np.ascontiguousarray(t, dtype=np.uint32)
        kpo = np.zeros(n_frames + 1, dtype=np.int64)
        kpo[1:] = np.cumsum(np.asarray(keypoints_per_frame, dtype=np.int64))
        h = C.c_void_p()
        ctx._check(ctx._L.eacham_graph_create(ctx.handle, n_frames, pairs.ctypes.data, pairs.shape[0], counts.ctypes.data, offsets.ctypes.data,
                                              q.ctypes.data, t.ctypes.data, kpo.ctypes.data, C.byref(h)))
        self._h = h

    def set_frame(self, frame: int, valid: bool, has3d=None):
        f = None if has3d is None else np.ascontiguousarray(has3d, dtype=np.uint8)
        self.ctx._check(self.ctx._L.eacham_graph_set_frame(self._h, int(frame), int(bool(valid)), None if f is None else f.ctypes.data,
                                                           0 if f is None else f.size))

    def set_frames(self, frames, valid, has3d_per_frame):
        """eacham_graph_set_frames: several frames in one copy + one kernel (has3d_per_frame[i]: the full flag array of frames[i])."""
        fr = np.ascontiguousarray(frames, dtype=np.int32)
        va = np.ascontiguousarray(valid, dtype=np.uint8)
        off = np.zeros(len(fr) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(a) for a in has3d_per_frame])
        fl = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.uint8) for a in has3d_per_frame]) if len(fr) else np.zeros(0, np.uint8))
        self.ctx._check(self.ctx._L.eacham_graph_set_frames(self._h, int(len(fr)), fr.ctypes.data, va.ctypes.data, fl.ctypes.data if fl.size else None, off.ctypes.data))

    def query(self, excluded_frames=()):
        ex = np.ascontiguousarray(list(excluded_frames), dtype=np.int32)
        best = np.zeros(3, dtype=np.uint32)
        self.ctx._check(self.ctx._L.eacham_graph_query(self._h, ex.ctypes.data if ex.size else None, int(ex.size), best.ctypes.data))
        return int(best[0]), int(best[1]), int(best[2])

    def close(self):
        if self._h:
            self.ctx._L.eacham_graph_destroy(self._h)
            self._h = None
