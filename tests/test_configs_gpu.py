"""GPU: the BASELINE.json configurations that are stand-ins for datasets absent from the image
(SURVEY.md §0 fact 9, §8(d) "Other configs as inputs"), at their FULL sizes:

  configs[2]  TUM fr1/desk  (config/ConfigTUM.json:3,28: <= 500 frames x 600 features)  -> all-pairs matching of
              500 x 600 x 128-D + a sequence of local-window RefineBA calls (apps/sfm/main.cpp:207)
  configs[4]  KITTI seq-00  (config/ConfigKITTI.json:3,29: 100 frames x 1500 features)   -> pair matching through
              the shard path (eacham_order_pairs / eacham_shard_bounds, one shard per rank, assembled match graph)

Parity: bit-exact indices against the CPU oracle on seeded samples the oracle finishes in seconds, plus
size-independent properties over the whole job; poses / points within 1e-5 relative for the BA windows.
The RCCL all-gather itself is exercised by bench.py --gpus N on a multi-GPU node and by the world-size-2 gloo
test (tests/test_shard_gloo.py); here the shards of an 8-rank run are produced one after the other on the one GPU.
"""
import functools

import numpy as np
import pytest

from eacham_amd import ba, shard, synth
import oracle_api as O

pytestmark = pytest.mark.gpu
POSE_POINT_RTOL = 1e-5  # north_star tolerance


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def _upload(ctx, descs):
    ctx.clear_descriptors()
    for f, d in enumerate(descs):
        ctx.upload_descriptors(f, d)


@pytest.fixture(scope="module")
def tum():
    sc = synth.make_scene(500, 30_000, 10, seed=3)          # 600 observed landmarks per frame
    descs, ids = synth.make_frame_descriptors(sc, 600, 128, seed=3)
    return sc, descs, ids


def test_config3_tum_standin_all_pairs(hip_ctx, tum):
    sc, descs, ids = tum
    pairs = shard.order_pairs(synth.all_pairs(500))         # 124 750 unordered pairs
    _upload(hip_ctx, descs)
    counts, offsets, q, t, stats = hip_ctx.match_all_pairs(pairs)
    assert counts.shape == (124_750,) and offsets[-1] == counts.sum() == len(q) == len(t)
    # (1) bit-exact against the oracle on a seeded sample of pairs: edges and non-edges alike
    rng = np.random.default_rng(3)
    edge_idx = np.nonzero(counts)[0]
    sample = np.unique(np.concatenate([rng.choice(edge_idx, 150, replace=False), rng.choice(len(pairs), 150, replace=False)]))
    want = O.match_all_pairs(descs, pairs[sample])
    assert np.array_equal(counts[sample], want[0]) and np.array_equal(stats[sample], want[4])
    got_q = np.concatenate([q[offsets[p]:offsets[p + 1]] for p in sample])
    got_t = np.concatenate([t[offsets[p]:offsets[p + 1]] for p in sample])
    assert np.array_equal(got_q, want[2]) and np.array_equal(got_t, want[3])
    # (2) size-independent properties of the whole job
    assert np.all(counts[counts > 0] > 30) and np.all(stats[:, 2] <= np.minimum(stats[:, 0], stats[:, 1]))  # thresholds, mutual <= directed
    assert np.array_equal(counts > 0, stats[:, 3] == 1)
    pid = np.repeat(np.arange(len(pairs)), counts)
    lq = np.array([ids[f] for f in range(500)])[pairs[pid, 0], q]
    lt = np.array([ids[f] for f in range(500)])[pairs[pid, 1], t]
    assert (lq >= 0).all() and np.mean(lq == lt) > 0.999    # a mutual match joins two observations of one landmark
    for p in rng.choice(edge_idx, 50, replace=False):       # sorted by q, and a matching: q and t both unique
        qq, tt = q[offsets[p]:offsets[p + 1]], t[offsets[p]:offsets[p + 1]]
        assert np.all(np.diff(qq.astype(np.int64)) > 0) and len(np.unique(tt)) == len(tt)
    # the helix geometry: frames more than a turn's worth of neighbours apart share nothing -> the graph is banded
    assert 2_000 < len(edge_idx) < 40_000
    # (3) swapping the roles of the two frames transposes the matches (one distance tile serves both directions)
    some = rng.choice(edge_idx, 40, replace=False)
    c2, o2, q2, t2, _ = hip_ctx.match_all_pairs(pairs[some][:, ::-1])
    for k, p in enumerate(some):
        a = sorted(zip(q[offsets[p]:offsets[p + 1]].tolist(), t[offsets[p]:offsets[p + 1]].tolist()))
        b = sorted(zip(t2[o2[k]:o2[k + 1]].tolist(), q2[o2[k]:o2[k + 1]].tolist()))
        assert a == b


@pytest.mark.parametrize("frame", [0, 100, 250, 499])
def test_config3_tum_standin_local_window_ba(hip_ctx, tum, frame):
    """RefineBA(currentFrameId, ..., refine_ba) on the window the reference would build around `frame`
    (BundleAdjuster.cpp:123-145): same LM trace as the oracle, poses and points within the north-star tolerance.
    Frame 0's window holds the fixed node; the others have no fixed camera (only the Huber pose priors)."""
    sc, _, _ = tum
    w = synth.local_window(sc, frame)
    A = ba.BaArrays.from_scene(w)
    assert A.cam_T_wc.shape[0] >= 10 and int(A.cam_fixed.sum()) == (1 if 0 in w["frames"] else 0)
    assert (np.bincount(A.obs_point, minlength=len(A.points)) >= 1).all() and (A.point_observers == 10).all()
    cfg = ba.OptimizerConfig.refine_ba()
    out = ba.RefineBA(hip_ctx, A, cfg)
    ref = O.ba_solve(A, cfg)
    assert out.status == ref.status == 0
    assert (out.outer_iterations, out.inner_iterations) == (ref.outer_iterations, ref.inner_iterations)
    assert np.array_equal(out.trace[:, 3:], ref.trace[:, 3:]) and np.allclose(out.trace[:, :2], ref.trace[:, :2], rtol=1e-6)
    assert rel(out.cam_T_wc, ref.cam_T_wc) < POSE_POINT_RTOL and rel(out.points, ref.points) < POSE_POINT_RTOL
    assert np.allclose(out.K, ref.K, rtol=1e-8)
    assert out.final_error < 0.1 * out.initial_error


def test_config3_tum_standin_window_sequence(hip_ctx, tum):
    """The incremental loop's pattern: one local window per added frame, each a new problem on the same context."""
    sc, _, _ = tum
    cfg = ba.OptimizerConfig.refine_ba()
    errs = []
    for f in range(200, 212):
        out = ba.RefineBA(hip_ctx, ba.BaArrays.from_scene(synth.local_window(sc, f)), cfg)
        assert out.status == 0 and 1 <= out.outer_iterations <= 30
        errs.append(out.final_error / out.initial_error)
    assert max(errs) < 0.1


def test_single_process_communicator_runs_the_rccl_gather(hip_ctx):
    """eacham_comm_init / eacham_match_all_pairs_sharded with the one device this box has:
    RCCL is loaded, ncclCommInitAll builds a 1-rank communicator, the two ncclAllGather calls run on the context's stream
    behind the matching, and the assembled graph — returned in the CALLER's pair order, not the train-frame order the
    shards use — equals eacham_match_all_pairs on the same frames."""
    sc = synth.make_scene(24, 4000, 8, seed=12)
    descs, _ = synth.make_frame_descriptors(sc, 700, 128, seed=12)
    descs[5] = descs[5][:333]
    pairs = synth.all_pairs(24)
    rng = np.random.default_rng(3)
    pairs = pairs[rng.permutation(len(pairs))]              # the caller's order is arbitrary
    _upload(hip_ctx, descs)
    want = hip_ctx.match_all_pairs(pairs)
    with shard.Comm(1) as comm:
        assert comm.size == 1
        for f, d in enumerate(descs):
            comm.upload_descriptors(f, d)
        got = comm.match_all_pairs(pairs)
        again = comm.match_all_pairs(pairs[:7])             # buffers are reused; a shorter list afterwards
    for g, w in zip(got, want[:4]):
        assert np.array_equal(g, w)
    assert want[0].sum() > 1000
    assert np.array_equal(again[0], want[0][:7]) and np.array_equal(again[2], want[2][:want[1][7]])
    with pytest.raises(Exception):
        shard.Comm(2)                                       # a second device does not exist on this box


def test_config5_kitti_standin_through_the_shard_path(hip_ctx):
    sc = synth.make_scene(100, 15_000, 10, seed=5)          # 1500 observed landmarks per frame
    descs, ids = synth.make_frame_descriptors(sc, 1500, 128, seed=5)
    pairs = shard.order_pairs(synth.all_pairs(100))         # 4950 pairs, ordered by train frame
    _upload(hip_ctx, descs)
    whole = hip_ctx.match_all_pairs(pairs)                  # the N = 1 shard is the whole job
    world = 8
    cap = shard.shard_capacity(len(pairs), world)
    parts = [hip_ctx.match_all_pairs(shard.shard_pairs(pairs, world, r)) for r in range(world)]
    edge_cap = max(int(p[0].sum()) for p in parts)
    g_counts = np.zeros((world, cap), np.int32)
    g_edges = np.zeros((world, edge_cap, 2), np.int32)
    for r, p in enumerate(parts):                           # what every rank contributes to the all-gather
        g_counts[r, :len(p[0])] = p[0]
        g_edges[r, :len(p[2]), 0], g_edges[r, :len(p[2]), 1] = p[2], p[3]
    counts, offsets, q, t = shard.assemble_match_graph(g_counts, g_edges, len(pairs), world, cap, edge_cap)
    assert np.array_equal(counts, whole[0]) and np.array_equal(offsets, whole[1])
    assert np.array_equal(q, whole[2]) and np.array_equal(t, whole[3])
    # bit-exact against the oracle on a sample of pairs
    rng = np.random.default_rng(5)
    sample = np.sort(rng.choice(len(pairs), 120, replace=False))
    want = O.match_all_pairs(descs, pairs[sample])
    assert np.array_equal(counts[sample], want[0])
    assert np.array_equal(np.concatenate([q[offsets[p]:offsets[p + 1]] for p in sample]), want[2])
    assert np.array_equal(np.concatenate([t[offsets[p]:offsets[p + 1]] for p in sample]), want[3])
    pid = np.repeat(np.arange(len(pairs)), counts)
    idm = np.array(ids)
    assert np.mean(idm[pairs[pid, 0], q] == idm[pairs[pid, 1], t]) > 0.999 and counts.sum() > 50_000


@functools.lru_cache(maxsize=2)
def _s200_descriptors(dim):
    return synth.make_frame_descriptors(synth.make_scene(200, 50_000, 10), 2000, dim)


def boundary_sample(starts_by_form, npairs, counts, rng, n_edges=160, n_any=160):
    """Indices of the pairs an oracle run must cover: the two pairs on either side of EVERY launch boundary of the job (both
    forms of the column direction cut it differently), the first and the last pair, and seeded edges / arbitrary pairs."""
    pick = {0, npairs - 1}
    for starts in starts_by_form:
        for s in starts:
            pick.update(int(s) + d for d in (-2, -1, 0, 1))
    edge_idx = np.nonzero(counts)[0]
    pick.update(rng.choice(edge_idx, min(n_edges, len(edge_idx)), replace=False).tolist())
    pick.update(rng.choice(npairs, min(n_any, npairs), replace=False).tolist())
    return np.array(sorted(p for p in pick if 0 <= p < npairs), dtype=np.int64), edge_idx


@pytest.mark.parametrize("frames,dim,min_batches", [(200, 256, 3), (200, 128, 2), (100, 256, 2)],
                         ids=["s200_d256_headline", "s200_d128", "config2_100x2000x256"])
def test_metric_scene_matching_job_against_the_oracle(hip_ctx, frames, dim, min_batches):
    """The job BASELINE.json's metric is quoted on, run as bench.py runs it — ONE eacham_match_all_pairs over ALL unordered pairs of
    the scene (apps/sfm/main.cpp:84-147 is one loop over all frames): S200 at 256-D (19 900 pairs: several launches, two workspace
    slots, the candidate pass on the second stream beside the next launch's sweep), the same at SIFT's 128-D, and configs[1]
    (100 frames). Bit-exact against the oracle on >= 300 pairs incl. both sides of every launch boundary; whole-job properties;
    two back-to-back runs bit for bit (slot reuse across calls)."""
    descs, ids = _s200_descriptors(dim)                       # the metric scene; configs[1] = its first 100 frames (as bench.py's c2)
    descs, ids = descs[:frames], ids[:frames]
    pairs = shard.order_pairs(synth.all_pairs(frames))
    _upload(hip_ctx, descs)
    lean_starts, lean_slots = hip_ctx.match_batches(len(pairs), stats=False)
    full_starts, _ = hip_ctx.match_batches(len(pairs), stats=True)
    assert len(lean_starts) >= min_batches and lean_slots == 2, (lean_starts, lean_slots)   # the path the headline runs
    counts, offsets, q, t, stats = hip_ctx.match_all_pairs(pairs)      # conftest: full-column form AND lean form, identical graphs
    assert offsets[-1] == counts.sum() == len(q) == len(t)
    rng = np.random.default_rng(frames + dim)
    sample, edge_idx = boundary_sample([lean_starts, full_starts], len(pairs), counts, rng)
    assert len(sample) >= 300
    want = O.match_all_pairs(descs, pairs[sample])
    assert np.array_equal(counts[sample], want[0]) and np.array_equal(stats[sample], want[4])
    assert np.array_equal(np.concatenate([q[offsets[p]:offsets[p + 1]] for p in sample]), want[2])
    assert np.array_equal(np.concatenate([t[offsets[p]:offsets[p + 1]] for p in sample]), want[3])
    # whole-job properties
    assert np.all(counts[counts > 0] > 30) and np.all(stats[:, 2] <= np.minimum(stats[:, 0], stats[:, 1]))
    assert np.array_equal(counts > 0, stats[:, 3] == 1)
    pid = np.repeat(np.arange(len(pairs)), counts)
    idm = np.array(ids)
    lq, lt = idm[pairs[pid, 0], q], idm[pairs[pid, 1], t]
    assert (lq >= 0).all() and np.mean(lq == lt) > 0.999        # a mutual match joins two observations of one landmark
    for p in rng.choice(edge_idx, 60, replace=False):           # sorted by q; q and t both unique
        qq, tt = q[offsets[p]:offsets[p + 1]], t[offsets[p]:offsets[p + 1]]
        assert np.all(np.diff(qq.astype(np.int64)) > 0) and len(np.unique(tt)) == len(tt)
    assert len(edge_idx) > frames                               # the helix: every frame has covisible neighbours
    # the lean form twice more, back to back (no host work in between: the second call's first launch reuses slot 0 behind the
    # first call's last events), bit for bit
    a = hip_ctx.match_all_pairs(pairs, stats=False)
    b = hip_ctx.match_all_pairs(pairs, stats=False)
    for x, y, w in zip(a[:4], b[:4], (counts, offsets, q, t)):
        assert np.array_equal(x, y) and np.array_equal(x, w)
    # the device-resident entry point bench.py times, two calls queued without a synchronisation between them
    import torch
    dev = torch.device("cuda", 0)
    ext = torch.cuda.ExternalStream(hip_ctx.stream, device=dev)
    with torch.cuda.stream(ext):
        pd = torch.from_numpy(pairs).to(dev)
        outs = []
        for _ in range(2):
            o = {"counts": torch.zeros(len(pairs), dtype=torch.int32, device=dev), "offsets": torch.zeros(len(pairs) + 1, dtype=torch.int64, device=dev),
                 "edges": torch.zeros(max(len(q), 1) * 2, dtype=torch.int32, device=dev), "total": torch.zeros(1, dtype=torch.int64, device=dev)}
            outs.append(o)
        hip_ctx.sync()
        for o in outs:
            hip_ctx.match_all_pairs_dev(pd.data_ptr(), len(pairs), o["counts"].data_ptr(), o["offsets"].data_ptr(), o["edges"].data_ptr(),
                                        len(q), o["total"].data_ptr())
        hip_ctx.sync()
    for o in outs:
        assert int(o["total"].item()) == len(q)
        assert np.array_equal(o["counts"].cpu().numpy(), counts) and np.array_equal(o["offsets"].cpu().numpy(), offsets)
        e = o["edges"].cpu().numpy().view(np.uint32).reshape(-1, 2)
        assert np.array_equal(e[:len(q), 0], q) and np.array_equal(e[:len(q), 1], t)
