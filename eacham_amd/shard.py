"""Multi-GPU sharding of the pair-matching stage.

The pair loop of apps/sfm/main.cpp:84-147 is embarrassingly parallel over unordered frame pairs.
One process per GPU: every rank holds all descriptors (S200: 102 MB int8), matches its contiguous
shard of the (train-frame-ordered) pair list, and the match graph is assembled on every rank by an
all-gather (RCCL over xGMI when the tensors live on GPUs; the same code runs on gloo/CPU tensors in
the tests). BA does not shard at these sizes ("replicas only").
"""
from __future__ import annotations

import numpy as np


def order_pairs(pairs: np.ndarray) -> np.ndarray:
    """Sorts pairs by train frame (then query frame): consecutive workgroups stream the same
    B-operand frame, which keeps it in the XCD's L2."""
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    return pairs[np.lexsort((pairs[:, 0], pairs[:, 1]))]


def shard_bounds(npairs: int, world: int) -> np.ndarray:
    """world+1 boundaries of the contiguous shards (sizes differ by at most one)."""
    base, rem = divmod(npairs, world)
    sizes = np.full(world, base, dtype=np.int64)
    sizes[:rem] += 1
    return np.concatenate([[0], np.cumsum(sizes)])


def pair_weights(pairs: np.ndarray, rows) -> np.ndarray:
    """Cost of each pair for the matcher: its distance matrix, rows(f1) * rows(f2) (`rows`: frame id -> row count)."""
    pairs = np.asarray(pairs).reshape(-1, 2)
    r = np.asarray([rows[int(f)] for f in range(int(pairs.max()) + 1)] if len(pairs) else [], dtype=np.int64)
    return r[pairs[:, 0]] * r[pairs[:, 1]] if len(pairs) else np.zeros(0, np.int64)


def shard_bounds_weighted(weights, world: int) -> np.ndarray:
    """world+1 boundaries of a WORK-balanced contiguous cut (mirror of eacham_shard_bounds_weighted): shard r starts at the
    smallest k whose prefix weight P[k] satisfies P[k] * world >= r * P[n]. Equal pair counts are the wrong cut for ragged
    frames (a pair costs rows(f1) * rows(f2)); with equal weights this is the count cut up to rounding."""
    w = np.asarray(weights, dtype=np.int64)
    n = len(w)
    if n == 0 or int(w.sum()) == 0:
        return shard_bounds(n, world)
    P = np.concatenate([[0], np.cumsum(w)])
    W = int(P[-1])
    b = np.empty(world + 1, dtype=np.int64)
    k = 0
    for r in range(world):
        while k < n and int(P[k]) * world < r * W:
            k += 1
        b[r] = k
    b[world] = n
    return b


def shard_pairs(pairs: np.ndarray, world: int, rank: int, bounds=None) -> np.ndarray:
    b = shard_bounds(len(pairs), world) if bounds is None else bounds
    return np.ascontiguousarray(pairs[b[rank]:b[rank + 1]])


def shard_capacity(npairs: int, world: int) -> int:
    return -(-npairs // world) if world > 0 else npairs


def all_gather_match_graph(counts, edges, shard_cap: int, edge_cap: int, world: int, g_counts=None, g_edges=None,
                           async_op: bool = False):
    """The collective step. counts: int32 [shard_cap] (this rank's per-pair counts, zero padded),
    edges: int32 [2*edge_cap] ({q, t} pairs, this rank's CSR payload, padded). Returns the gathered
    (world*shard_cap, world*2*edge_cap) tensors; preallocate g_* to keep the step allocation-free."""
    import torch
    import torch.distributed as dist
    if g_counts is None:
        g_counts = torch.empty(world * shard_cap, dtype=counts.dtype, device=counts.device)
    if g_edges is None:
        g_edges = torch.empty(world * 2 * edge_cap, dtype=edges.dtype, device=edges.device)
    if async_op:  # the caller overlaps the exchange with the next batch of matching and waits before reusing the buffers
        return g_counts, g_edges, [dist.all_gather_into_tensor(g_counts, counts, async_op=True),
                                   dist.all_gather_into_tensor(g_edges, edges, async_op=True)]
    dist.all_gather_into_tensor(g_counts, counts)
    dist.all_gather_into_tensor(g_edges, edges)
    return g_counts, g_edges


def assemble_match_graph(g_counts: np.ndarray, g_edges: np.ndarray, npairs: int, world: int, shard_cap: int, edge_cap: int, bounds=None):
    """Host-side view of the gathered buffers as one CSR over the (ordered) pair list:
    (counts[npairs], offsets[npairs+1], q, t). `bounds`: the shard boundaries used (default: the equal-count cut)."""
    b = shard_bounds(npairs, world) if bounds is None else bounds
    g_counts = np.asarray(g_counts).reshape(world, shard_cap)
    g_edges = np.asarray(g_edges).reshape(world, edge_cap, 2)
    counts, qs, ts = [], [], []
    for r in range(world):
        c = g_counts[r, : b[r + 1] - b[r]]
        n = int(c.sum())
        counts.append(c)
        qs.append(g_edges[r, :n, 0])
        ts.append(g_edges[r, :n, 1])
    counts = np.concatenate(counts).astype(np.int32)
    offsets = np.zeros(npairs + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    return counts, offsets, np.concatenate(qs).astype(np.uint32), np.concatenate(ts).astype(np.uint32)


class Comm:
    """Single-process multi-GPU communicator of the C-ABI (eacham_comm_init / eacham_match_all_pairs_sharded): one
    context + one host thread per device, descriptors replicated, RCCL all-gather of the match graph. Used by the
    tests and by callers that run the reference's one-process layout (apps/sfm/main.cpp:31)."""

    def __init__(self, ndev: int = 1, devices=None):
        import ctypes as C
        from . import capi
        self._C, self._capi, self._L = C, capi, capi.lib()
        h = C.c_void_p()
        dev = None if devices is None else np.ascontiguousarray(devices, dtype=np.int32)
        rc = self._L.eacham_comm_init(int(ndev), None if dev is None else dev.ctypes.data, C.byref(h))
        if rc:
            raise capi.EachamError(rc, "eacham_comm_init failed (no device, RCCL not loadable, or ncclCommInitAll refused the devices)")
        self.handle, self.size = h, int(self._L.eacham_comm_size(h))
        self.rows = {}

    def _check(self, rc):
        if rc:
            raise self._capi.EachamError(rc, (self._L.eacham_comm_last_error(self.handle) or b"").decode())

    def upload_descriptors(self, frame_id: int, desc: np.ndarray):
        d = np.ascontiguousarray(desc, dtype=np.float32)
        self._check(self._L.eacham_comm_upload_descriptors(self.handle, int(frame_id), d.ctypes.data, d.shape[0], d.shape[1] if d.ndim == 2 else 0))
        self.rows[int(frame_id)] = d.shape[0]

    def match_all_pairs(self, pairs: np.ndarray, ratio: float = 0.8, min_dir: int = 30, min_mutual: int = 30):
        """(counts, offsets, q, t) in the caller's pair order, as HipContext.match_all_pairs returns them."""
        C = self._C
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        n = len(pairs)
        cap = int(sum(self.rows.get(int(f), 0) for f in pairs[:, 0])) + 1
        counts = np.zeros(n, np.int32); offsets = np.zeros(n + 1, np.int64)
        q = np.zeros(cap, np.uint32); t = np.zeros(cap, np.uint32)
        total = C.c_int64(0)
        self._check(self._L.eacham_match_all_pairs_sharded(self.handle, pairs.ctypes.data, n, float(ratio), int(min_dir), int(min_mutual),
                                                           counts.ctypes.data, offsets.ctypes.data, q.ctypes.data, t.ctypes.data, cap, C.byref(total)))
        return counts, offsets, q[: total.value].copy(), t[: total.value].copy()

    def match_run(self, pairs: np.ndarray, ratio: float = 0.8, min_dir: int = 30, min_mutual: int = 30, balance: bool = True) -> int:
        """eacham_comm_match_run: match + all-gather, the graph stays resident on every device; returns its match count."""
        C = self._C
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        total = C.c_int64(0)
        self._check(self._L.eacham_comm_match_run(self.handle, pairs.ctypes.data, len(pairs), float(ratio), int(min_dir), int(min_mutual),
                                                  int(bool(balance)), C.byref(total)))
        self._last = pairs
        return int(total.value)

    def match_fetch(self):
        """eacham_comm_match_fetch: the graph of the last match_run as (counts, offsets, q, t) in the caller's pair order."""
        C = self._C
        n = len(self._last)
        cap = int(sum(self.rows.get(int(f), 0) for f in self._last[:, 0])) + 1
        counts = np.zeros(n, np.int32); offsets = np.zeros(n + 1, np.int64)
        q = np.zeros(cap, np.uint32); t = np.zeros(cap, np.uint32)
        total = C.c_int64(0)
        self._check(self._L.eacham_comm_match_fetch(self.handle, counts.ctypes.data, offsets.ctypes.data, q.ctypes.data, t.ctypes.data, cap, C.byref(total)))
        return counts, offsets, q[: total.value].copy(), t[: total.value].copy()

    def ctx_handle(self, rank: int):
        """The eacham_ctx of device `rank` (owned by the communicator)."""
        return self._C.c_void_p(self._L.eacham_comm_ctx(self.handle, int(rank)))

    def close(self):
        if self.handle:
            self._L.eacham_comm_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
