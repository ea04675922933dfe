"""Host-side mirror of the reference matcher interface, on top of the C-ABI.

  FeatureMatcherHip.Match(d1, d2)    <->  FeatureMatcherFlann::Match / IFeatureMatcher<T>::Match
                                          (modules/base/features/FeatureMatcherFlann.h:14-19,
                                           modules/base/features/IFeatureMatcher.h:8-20)
  HipContext.match_all_pairs(pairs)  <->  the pair loop + mutual check of apps/sfm/main.cpp:84-147

Python is only the test/bench driver here (the reference is C++; its adapter is
include/eacham/FeatureMatcherHip.hpp). Same names, argument meaning and error behaviour.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

# literals of the reference
RATIO = 0.8        # FeatureMatcherFlann.cpp:23 (the ctor's inliersRatio is stored but never used)
MIN_DIRECTED = 30  # apps/sfm/main.cpp:111  `matches12.size() < 30` -> drop
MIN_MUTUAL = 30    # apps/sfm/main.cpp:142  `bestMatches12.size() > 30` -> connect


class HipContext:
    """One HIP device + stream + resident descriptor store (eacham_ctx)."""

    def __init__(self, device_id: int = 0):
        self._L = capi.lib()
        h = C.c_void_p()
        rc = self._L.eacham_ctx_create(device_id, C.byref(h))
        if rc != capi.OK:
            raise capi.EachamError(rc, "eacham_ctx_create failed (no HIP device? the hot path has no CPU fallback)")
        self._h = h

    @classmethod
    def borrowed(cls, handle):
        """A view of a context that somebody else owns (a device of an eacham_comm): never destroyed from here."""
        self = cls.__new__(cls)
        self._L = capi.lib()
        self._h = handle
        self._borrowed = True
        return self

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                self._L.eacham_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc: int):
        if rc != capi.OK:
            raise capi.EachamError(rc, self._L.eacham_last_error(self._h).decode())

    @property
    def handle(self):
        return self._h

    @property
    def stream(self) -> int:
        return int(self._L.eacham_ctx_stream(self._h) or 0)

    def stream2_info(self) -> dict:
        """What the second-stream search of eacham_ctx_create decided (eacham_ctx_stream2_info)."""
        a, ms = C.c_int(-1), C.c_float(-1.0)
        self._check(self._L.eacham_ctx_stream2_info(self._h, C.byref(a), C.byref(ms)))
        return {"attempt": a.value, "lead_ms": round(ms.value, 4), "own_queue": bool(ms.value > 0.010)}

    def sync(self):
        self._check(self._L.eacham_ctx_sync(self._h))

    # ---- descriptor store ------------------------------------------------------------------
    def upload_descriptors(self, frame_id: int, desc: np.ndarray):
        d = np.ascontiguousarray(desc, dtype=np.float32)
        if d.ndim != 2:
            raise ValueError("descriptors must be an N x D matrix")
        self._check(self._L.eacham_upload_descriptors(self._h, frame_id, d.ctypes.data, d.shape[0], d.shape[1]))

    def upload_descriptors_f32(self, frame_id: int, desc: np.ndarray):
        """Float descriptors (SuperPoint / LightGlue style): fp32 MFMA path, any values."""
        d = np.ascontiguousarray(desc, dtype=np.float32)
        if d.ndim != 2:
            raise ValueError("descriptors must be an N x D matrix")
        self._check(self._L.eacham_upload_descriptors_f32(self._h, frame_id, d.ctypes.data, d.shape[0], d.shape[1]))

    def upload_descriptors_dev(self, frame_id: int, dev_ptr: int, n: int, dim: int):
        self._check(self._L.eacham_upload_descriptors_dev(self._h, frame_id, C.c_void_p(dev_ptr), n, dim))

    def frame_rows(self, frame_id: int) -> int:
        n = self._L.eacham_frame_rows(self._h, frame_id)
        if n < 0:
            self._check(n)
        return n

    def clear_descriptors(self):
        self._check(self._L.eacham_clear_descriptors(self._h))

    # ---- matching --------------------------------------------------------------------------
    def match_pair(self, f1: int, f2: int, ratio: float = RATIO):
        cap = max(self.frame_rows(f1), 1)
        q = np.empty(cap, dtype=np.uint32)
        t = np.empty(cap, dtype=np.uint32)
        cnt = C.c_int(0)
        self._check(self._L.eacham_match_pair(self._h, f1, f2, ratio, q.ctypes.data, t.ctypes.data, cap, C.byref(cnt)))
        return q[:cnt.value].copy(), t[:cnt.value].copy()

    def match_all_pairs(self, pairs: np.ndarray, ratio: float = RATIO, min_dir: int = MIN_DIRECTED,
                        min_mutual: int = MIN_MUTUAL, cap: int | None = None, stats: bool = True):
        """Returns (counts, offsets, q, t, stats): CSR over pairs, see include/eacham_hip.h. stats=False passes NULL for the
        per-pair statistics (what the pair loop of apps/sfm/main.cpp needs: the library then evaluates the column direction
        for candidate columns only) and returns None in their place."""
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        npairs = pairs.shape[0]
        if cap is None:
            cap = int(sum(self.frame_rows(int(p[0])) for p in pairs)) if npairs else 0
        counts = np.zeros(npairs, dtype=np.int32)
        offsets = np.zeros(npairs + 1, dtype=np.int64)
        q = np.empty(max(cap, 1), dtype=np.uint32)
        t = np.empty(max(cap, 1), dtype=np.uint32)
        st = np.zeros((npairs, 4), dtype=np.int32) if stats else None
        total = C.c_int64(0)
        self._check(self._L.eacham_match_all_pairs(
            self._h, pairs.ctypes.data, npairs, ratio, min_dir, min_mutual, counts.ctypes.data,
            offsets.ctypes.data, q.ctypes.data, t.ctypes.data, cap, C.byref(total), st.ctypes.data if stats else None))
        return counts, offsets, q[:total.value].copy(), t[:total.value].copy(), st

    def match_batches(self, npairs: int, stats: bool = False):
        """(starts, slots): first pair of every launch the library would cut a job of `npairs` pairs into, and the number of
        workspace slots the launches rotate through (eacham_match_debug_batches)."""
        starts = np.zeros(4096, dtype=np.int32)
        nb, ns = C.c_int(0), C.c_int(0)
        self._check(self._L.eacham_match_debug_batches(self._h, npairs, int(stats), starts.ctypes.data, len(starts), C.byref(nb), C.byref(ns)))
        return starts[:min(nb.value, len(starts))].copy(), ns.value

    def match_pairs_directed(self, frames, ordered_pairs, ratio: float = RATIO, f32: bool = False) -> list:
        """Uploads `frames` (list of N x D matrices) as frames 0.. and runs every ordered pair (i, j) as one directed
        Match(frames[i], frames[j]) in ONE launch sequence; returns a list of {queryIdx: trainIdx} dicts."""
        self.clear_descriptors()
        for f, d in enumerate(frames):
            (self.upload_descriptors_f32 if f32 else self.upload_descriptors)(f, d)
        pairs = np.ascontiguousarray(ordered_pairs, dtype=np.int32).reshape(-1, 2)
        npairs = pairs.shape[0]
        cap = int(sum(frames[int(p[0])].shape[0] for p in pairs))
        counts = np.zeros(npairs, dtype=np.int32)
        offsets = np.zeros(npairs + 1, dtype=np.int64)
        q = np.empty(max(cap, 1), dtype=np.uint32)
        t = np.empty(max(cap, 1), dtype=np.uint32)
        total = C.c_int64(0)
        self._check(self._L.eacham_match_pairs_directed(self._h, pairs.ctypes.data, npairs, ratio, counts.ctypes.data,
                                                        offsets.ctypes.data, q.ctypes.data, t.ctypes.data, cap, C.byref(total)))
        return [dict(zip(q[offsets[p]:offsets[p + 1]].tolist(), t[offsets[p]:offsets[p + 1]].tolist())) for p in range(npairs)]

    def match_all_pairs_dev(self, pairs_dev: int, npairs: int, counts_dev: int, offsets_dev: int,
                            edges_dev: int, edge_cap: int, total_dev: int, stats_dev: int = 0,
                            ratio: float = RATIO, min_dir: int = MIN_DIRECTED, min_mutual: int = MIN_MUTUAL):
        vp = C.c_void_p
        self._check(self._L.eacham_match_all_pairs_dev(
            self._h, vp(pairs_dev), npairs, ratio, min_dir, min_mutual, vp(counts_dev), vp(offsets_dev),
            vp(edges_dev), edge_cap, vp(total_dev), vp(stats_dev) if stats_dev else None))

    # ---- profiling -------------------------------------------------------------------------
    def profile_enable(self, on: bool = True):
        self._check(self._L.eacham_profile_enable(self._h, int(on)))

    def profile_reset(self):
        self._check(self._L.eacham_profile_reset(self._h))

    def profile_get(self, kernel_id: int):
        n, ms = C.c_int64(0), C.c_double(0.0)
        self._check(self._L.eacham_profile_get(self._h, kernel_id, C.byref(n), C.byref(ms)))
        return n.value, ms.value


class FeatureMatcherHip:
    """Drop-in shape of eacham::FeatureMatcherFlann (FeatureMatcherFlann.h:11-25).

    `Match(descriptor1, descriptor2)` takes two N x D fp32 matrices (cv::Mat CV_32F layout) and
    returns {queryIdx: trainIdx}. As in the reference, the constructor's `inliersRatio` is kept
    but the ratio test uses the literal 0.8 unless `ratio` is given explicitly.
    """

    def __init__(self, inliersRatio: float = 0.8, ratio: float = RATIO, context: HipContext | None = None):
        self.inliersRatio = inliersRatio
        self.ratio = ratio
        self.ctx = context or HipContext()
        self._f32 = False  # the first non-integer frame switches the instance to the fp32 path for good

    def Match(self, descriptor1: np.ndarray, descriptor2: np.ndarray) -> dict:
        """Two frame slots (0 and 1) of the context's store are rewritten per call; the store is cleared only
        when the descriptor kind changes. (The C++ adapter include/eacham/FeatureMatcherHip.hpp additionally
        caches uploads by buffer address and combines concurrent callers; numpy temporaries reuse addresses too
        freely for that to be safe here, and this class only drives tests.)"""
        if not self._f32:
            try:  # SIFT-style integers: exact int8 path
                self.ctx.upload_descriptors(0, descriptor1)
                self.ctx.upload_descriptors(1, descriptor2)
            except capi.EachamError as e:
                if e.code not in (capi.ERR_NOT_INTEGER, capi.ERR_UNSUPPORTED):
                    raise
                self._f32 = True  # other floats: fp32 MFMA path; all resident frames must be of one kind
                self.ctx.clear_descriptors()
        if self._f32:
            self.ctx.upload_descriptors_f32(0, descriptor1)
            self.ctx.upload_descriptors_f32(1, descriptor2)
        q, t = self.ctx.match_pair(0, 1, self.ratio)
        return dict(zip(q.tolist(), t.tolist()))

    def MatchPairs(self, frames, ordered_pairs) -> list:
        """eacham_match_pairs_directed: every (i, j) of `ordered_pairs` is one Match(frames[i], frames[j])."""
        return self.ctx.match_pairs_directed(frames, ordered_pairs, self.ratio, f32=self._f32)
