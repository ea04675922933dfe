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


def shard_pairs(pairs: np.ndarray, world: int, rank: int) -> np.ndarray:
    b = shard_bounds(len(pairs), world)
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


def assemble_match_graph(g_counts: np.ndarray, g_edges: np.ndarray, npairs: int, world: int, shard_cap: int, edge_cap: int):
    """Host-side view of the gathered buffers as one CSR over the (ordered) pair list:
    (counts[npairs], offsets[npairs+1], q, t)."""
    b = shard_bounds(npairs, world)
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
