"""CPU: the N > 1 path — pair sharding + all-gather of the match graph — with world_size 2 on gloo.
Each rank matches its shard with the CPU oracle (a stand-in for the device matcher, test only);
the collective and the assembly are the product code of eacham_amd/shard.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from eacham_amd import shard, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_everything_once():
    for n in [0, 1, 7, 19900]:
        for w in [1, 2, 3, 8]:
            b = shard.shard_bounds(n, w)
            assert b[0] == 0 and b[-1] == n and np.all(np.diff(b) >= 0) and np.diff(b).max() - np.diff(b).min() <= 1
            assert shard.shard_capacity(n, w) >= np.diff(b).max()
    pairs = synth.all_pairs(9)
    ordered = shard.order_pairs(pairs)
    assert sorted(map(tuple, ordered.tolist())) == sorted(map(tuple, pairs.tolist()))
    assert np.all(np.diff(ordered[:, 1]) >= 0)  # grouped by train frame
    got = np.concatenate([shard.shard_pairs(ordered, 4, r) for r in range(4)])
    assert np.array_equal(got, ordered)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, descs, pairs, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.shard_pairs(pairs, world, rank)
    c, o, q, t, st, _ = O.match_all_pairs(descs, mine, min_dir=3, min_mutual=2, nthreads=1)
    cap = shard.shard_capacity(len(pairs), world)
    tot = torch.tensor([len(q)], dtype=torch.int64)
    dist.all_reduce(tot, op=dist.ReduceOp.MAX)
    edge_cap = max(int(tot.item()), 1)
    counts = torch.zeros(cap, dtype=torch.int32)
    counts[: len(c)] = torch.from_numpy(c)
    edges = torch.zeros(2 * edge_cap, dtype=torch.int32)
    edges[: 2 * len(q)] = torch.from_numpy(np.stack([q, t], 1).astype(np.int32).reshape(-1))
    g_counts, g_edges = shard.all_gather_match_graph(counts, edges, cap, edge_cap, world)
    res = shard.assemble_match_graph(g_counts.numpy(), g_edges.numpy(), len(pairs), world, cap, edge_cap)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), counts=res[0], offsets=res[1], q=res[2], t=res[3])
    dist.barrier()
    dist.destroy_process_group()


def test_c_abi_sharding_helpers_agree_with_the_python_mirror():
    """eacham_order_pairs / eacham_shard_bounds (host-side, no device) against eacham_amd/shard.py."""
    import ctypes as C
    from eacham_amd import capi
    L = capi.lib()
    rng = np.random.default_rng(7)
    pairs = rng.integers(0, 40, size=(997, 2)).astype(np.int32)
    mine = np.ascontiguousarray(pairs.copy())
    assert L.eacham_order_pairs(mine.ctypes.data, len(mine)) == 0
    assert np.array_equal(mine, shard.order_pairs(pairs))
    for world in (1, 2, 3, 8, 1000):
        b = shard.shard_bounds(len(pairs), world)
        for rank in range(world):
            lo, hi = C.c_int32(-1), C.c_int32(-1)
            assert L.eacham_shard_bounds(len(pairs), world, rank, C.byref(lo), C.byref(hi)) == 0
            assert (lo.value, hi.value) == (b[rank], b[rank + 1])
    lo, hi = C.c_int32(), C.c_int32()
    assert L.eacham_shard_bounds(10, 0, 0, C.byref(lo), C.byref(hi)) < 0
    assert L.eacham_shard_bounds(10, 2, 2, C.byref(lo), C.byref(hi)) < 0
    assert L.eacham_order_pairs(None, 0) == 0


def test_c_abi_assembly_of_gathered_shards_agrees_with_the_python_mirror():
    """eacham_assemble_match_graph — the host half of eacham_match_all_pairs_sharded (what follows its RCCL all-gather;
    no device needed) — against eacham_amd/shard.py on synthetic gathered buffers, for worlds of 1 .. 8 ranks, with
    the caller's pair order restored from the train-frame order."""
    import ctypes as C
    from eacham_amd import capi
    L = capi.lib()
    rng = np.random.default_rng(11)
    for npairs, world in [(1, 1), (5, 8), (37, 2), (300, 3), (300, 8)]:
        pairs = rng.integers(0, 25, size=(npairs, 2)).astype(np.int32)
        order = np.lexsort((pairs[:, 0], pairs[:, 1])).astype(np.int32)  # sorted pair k sits at order[k] in the caller's list
        counts_sorted = rng.integers(0, 6, size=npairs).astype(np.int32)
        cap = shard.shard_capacity(npairs, world)
        b = shard.shard_bounds(npairs, world)
        edge_cap = max(1, max(int(counts_sorted[b[r]:b[r + 1]].sum()) for r in range(world)))
        g_counts = np.zeros((world, cap), np.int32)
        g_edges = rng.integers(0, 1 << 20, size=(world, edge_cap, 2)).astype(np.uint32)
        for r in range(world):
            g_counts[r, : b[r + 1] - b[r]] = counts_sorted[b[r]:b[r + 1]]
        want = shard.assemble_match_graph(g_counts, g_edges.astype(np.int64), npairs, world, cap, edge_cap)  # CSR over the SORTED list
        counts = np.zeros(npairs, np.int32); offsets = np.zeros(npairs + 1, np.int64)
        tot_cap = int(counts_sorted.sum())
        q = np.zeros(max(tot_cap, 1), np.uint32); t = np.zeros(max(tot_cap, 1), np.uint32)
        total = C.c_int64(-1)
        rc = L.eacham_assemble_match_graph(g_counts.ctypes.data, g_edges.ctypes.data, npairs, world, cap, edge_cap, order.ctypes.data,
                                           counts.ctypes.data, offsets.ctypes.data, q.ctypes.data, t.ctypes.data, tot_cap, C.byref(total))
        assert rc == 0 and total.value == tot_cap
        for k in range(npairs):   # sorted pair k -> caller's position order[k]
            p = int(order[k])
            assert counts[p] == want[0][k]
            assert np.array_equal(q[offsets[p]:offsets[p + 1]], want[2][want[1][k]:want[1][k + 1]])
            assert np.array_equal(t[offsets[p]:offsets[p + 1]], want[3][want[1][k]:want[1][k + 1]])
        if tot_cap > 0:  # a short output buffer is an error, the total is still reported
            rc = L.eacham_assemble_match_graph(g_counts.ctypes.data, g_edges.ctypes.data, npairs, world, cap, edge_cap, order.ctypes.data,
                                               counts.ctypes.data, offsets.ctypes.data, q.ctypes.data, t.ctypes.data, tot_cap - 1, C.byref(total))
            assert rc == capi.ERR_CAPACITY and total.value == tot_cap


def test_weighted_cut_balances_ragged_frames_and_agrees_with_the_c_abi():
    """eacham_shard_bounds_weighted / shard.shard_bounds_weighted: a pair costs rows(f1) * rows(f2) (apps/sfm/main.cpp:98-109
    hands every pair to whichever thread is free; a static cut has to weigh them). Ragged frames — 333 .. 1500 rows, some
    empty — in one job: every shard's work within one pair of the mean, where the equal-count cut is off by tens of per cent."""
    import ctypes as C
    from eacham_amd import capi
    L = capi.lib()
    rng = np.random.default_rng(5)
    rows = rng.integers(333, 1501, size=60)
    rows[[7, 31]] = 0                                       # empty frames: zero-weight pairs
    pairs = shard.order_pairs(synth.all_pairs(60))
    w = shard.pair_weights(pairs, rows)
    assert w.dtype == np.int64 and len(w) == len(pairs) and (w[(pairs == 7).any(1)] == 0).all()
    for world in (1, 2, 3, 8):
        b = shard.shard_bounds_weighted(w, world)
        got = np.zeros(world + 1, np.int32)
        assert L.eacham_shard_bounds_weighted(len(pairs), world, w.ctypes.data, got.ctypes.data) == 0
        assert np.array_equal(got, b)
        assert b[0] == 0 and b[-1] == len(pairs) and np.all(np.diff(b) >= 0)
        work = np.array([w[b[r]:b[r + 1]].sum() for r in range(world)])
        assert np.abs(work - w.sum() / world).max() <= w.max()             # within one pair of the mean
        if world == 8:
            bc = shard.shard_bounds(len(pairs), world)
            count_work = np.array([w[bc[r]:bc[r + 1]].sum() for r in range(world)])
            assert count_work.max() / (w.sum() / world) > 1.05 > work.max() / (w.sum() / world)   # what the count cut costs here
    # no weights / all-zero weights: the equal-count cut; bad input refused
    got = np.zeros(4, np.int32)
    assert L.eacham_shard_bounds_weighted(10, 3, None, got.ctypes.data) == 0 and np.array_equal(got, shard.shard_bounds(10, 3))
    z = np.zeros(10, np.int64)
    assert L.eacham_shard_bounds_weighted(10, 3, z.ctypes.data, got.ctypes.data) == 0 and np.array_equal(got, shard.shard_bounds(10, 3))
    assert np.array_equal(shard.shard_bounds_weighted(z, 3), shard.shard_bounds(10, 3))
    neg = np.array([1, -1, 1], np.int64)
    assert L.eacham_shard_bounds_weighted(3, 2, neg.ctypes.data, got.ctypes.data) == capi.ERR_INVALID
    assert L.eacham_shard_bounds_weighted(3, 0, None, got.ctypes.data) == capi.ERR_INVALID


def test_assembly_with_explicit_bounds_and_uneven_or_empty_shards():
    """eacham_assemble_match_graph_bounds on the shards a weighted cut produces — uneven, and EMPTY when there are fewer
    pairs than ranks — and eacham_comm_edge_region, the sizing rule of the all-gather's send buffers: every rank sends
    the same number of edge slots (the largest exact total), so every rank's region holds the largest shard bound; a region
    sized by the rank's own bound (8 bytes for an empty shard) was read out of bounds by the collective (round-3 advice)."""
    import ctypes as C
    from eacham_amd import capi
    L = capi.lib()
    rng = np.random.default_rng(13)
    for npairs, world in [(3, 8), (1, 2), (40, 4), (200, 8)]:
        pairs = rng.integers(0, 12, size=(npairs, 2)).astype(np.int32)
        order = np.lexsort((pairs[:, 0], pairs[:, 1])).astype(np.int32)
        w = rng.integers(1, 1000, size=npairs).astype(np.int64) ** 2
        b = shard.shard_bounds_weighted(w, world).astype(np.int32)
        assert npairs >= world or (np.diff(b) == 0).any()                                  # empty shards exist when pairs < ranks
        counts_sorted = rng.integers(0, 6, size=npairs).astype(np.int32)
        cap = max(1, int(np.diff(b).max()))
        edge_cap = max(1, max(int(counts_sorted[b[r]:b[r + 1]].sum()) for r in range(world)))
        g_counts = np.zeros((world, cap), np.int32)
        g_edges = rng.integers(0, 1 << 20, size=(world, edge_cap, 2)).astype(np.uint32)
        for r in range(world):
            g_counts[r, : b[r + 1] - b[r]] = counts_sorted[b[r]:b[r + 1]]
        want = shard.assemble_match_graph(g_counts, g_edges.astype(np.int64), npairs, world, cap, edge_cap, bounds=b)
        counts = np.zeros(npairs, np.int32); offsets = np.zeros(npairs + 1, np.int64)
        tot_cap = int(counts_sorted.sum())
        q = np.zeros(max(tot_cap, 1), np.uint32); t = np.zeros(max(tot_cap, 1), np.uint32)
        total = C.c_int64(-1)
        rc = L.eacham_assemble_match_graph_bounds(g_counts.ctypes.data, g_edges.ctypes.data, npairs, world, cap, edge_cap, b.ctypes.data,
                                                  order.ctypes.data, counts.ctypes.data, offsets.ctypes.data, q.ctypes.data, t.ctypes.data,
                                                  tot_cap, C.byref(total))
        assert rc == 0 and total.value == tot_cap
        for k in range(npairs):
            p = int(order[k])
            assert counts[p] == want[0][k]
            assert np.array_equal(q[offsets[p]:offsets[p + 1]], want[2][want[1][k]:want[1][k + 1]])
        bad = b.copy(); bad[-1] += 1
        assert L.eacham_assemble_match_graph_bounds(g_counts.ctypes.data, g_edges.ctypes.data, npairs, world, cap, edge_cap, bad.ctypes.data,
                                                    order.ctypes.data, counts.ctypes.data, offsets.ctypes.data, q.ctypes.data, t.ctypes.data,
                                                    tot_cap, C.byref(total)) == capi.ERR_INVALID
        # the send region of EVERY rank = the largest bound (an empty shard's own bound is 1)
        bound = np.array([1 + int(counts_sorted[b[r]:b[r + 1]].sum()) * 7 for r in range(world)], np.int64)
        region = C.c_int64(0)
        assert L.eacham_comm_edge_region(world, bound.ctypes.data, C.byref(region)) == 0
        assert region.value == bound.max() >= edge_cap
    assert L.eacham_comm_edge_region(0, None, C.byref(region)) == capi.ERR_INVALID


def test_communicator_needs_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from eacham_amd import capi
    with pytest.raises(capi.EachamError):
        shard.Comm(1)


def test_two_rank_all_gather_reproduces_the_single_process_graph(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api as O
    sc = synth.make_scene(6, 300, 3, seed=8)
    descs, _ = synth.make_frame_descriptors(sc, 96, 64, seed=8)
    descs[4] = descs[4][:50]
    pairs = shard.order_pairs(synth.all_pairs(6))
    want = O.match_all_pairs(descs, pairs, min_dir=3, min_mutual=2, nthreads=1)
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), descs, pairs, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):  # every rank ends up with the complete graph
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        assert np.array_equal(got["counts"], want[0]) and np.array_equal(got["offsets"], want[1])
        assert np.array_equal(got["q"], want[2]) and np.array_equal(got["t"], want[3])
    assert want[0].sum() > 0
