"""GPU: eacham_graph_best_pair through the C-ABI against the oracle."""
import numpy as np
import pytest

import oracle_api as O
from eacham_amd import HipContext, capi, synth
from eacham_amd import graph as G
from test_graph_oracle import scenario

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    with HipContext(0) as c:
        yield c


@pytest.mark.parametrize("n_frames,seed", [(5, 0), (9, 1), (17, 2), (40, 3), (64, 4)])
def test_matches_oracle(ctx, n_frames, seed):
    pairs, counts, offsets, q, t, valid, has3d, excluded = scenario(n_frames, seed)
    for ex in (None, excluded):
        got, ec = G.best_pair_for_valid(ctx, n_frames, pairs, counts, offsets, q, t, valid, has3d, ex, want_edge_counts=True)
        want, wec = O.graph_best_pair(n_frames, pairs, counts, offsets, q, t, valid, has3d, ex)
        assert got == want and np.array_equal(ec, wec)


@pytest.mark.parametrize("n_frames,seed", [(9, 1), (40, 3), (64, 4)])
def test_resident_graph_follows_the_state_frame_by_frame(ctx, n_frames, seed):
    """eacham_graph_create / _set_frame / _query: the graph uploaded once, the loop's state changes applied frame by frame —
    after every change the query must answer what the one-shot entry point (and the oracle) answer on the full state."""
    pairs, counts, offsets, q, t, valid, has3d, excluded = scenario(n_frames, seed)
    rng = np.random.default_rng(seed)
    rg = G.ResidentGraph(ctx, n_frames, pairs, counts, offsets, q, t, [len(a) for a in has3d])
    try:
        state_valid = np.zeros(n_frames, np.uint8)
        state_h = [np.zeros(len(a), np.uint8) for a in has3d]
        assert rg.query() == (G.NONE, G.NONE, 0)                       # nothing valid yet
        for f in rng.permutation(n_frames):                            # the frames arrive one by one
            state_valid[f] = valid[f]
            state_h[f] = np.asarray(has3d[f], np.uint8)
            rg.set_frame(int(f), bool(valid[f]), state_h[f])
            ex = [int(e) for e in np.flatnonzero(excluded)] if f % 2 else []
            exm = np.zeros(n_frames, np.uint8); exm[ex] = 1
            want, _ = O.graph_best_pair(n_frames, pairs, counts, offsets, q, t, state_valid, state_h, exm)
            assert rg.query(ex) == want == G.best_pair_for_valid(ctx, n_frames, pairs, counts, offsets, q, t, state_valid, state_h, exm)
        rg.set_frame(0, True)                                           # validity alone (flags untouched)
        state_valid[0] = 1
        want, _ = O.graph_best_pair(n_frames, pairs, counts, offsets, q, t, state_valid, state_h, None)
        assert rg.query() == want
        # the batched form (eacham_graph_set_frames): a few frames changed at once, as after a frame of the loop
        group = [int(f) for f in rng.permutation(n_frames)[:min(7, n_frames)]]
        for f in group:
            state_valid[f] ^= 1
            state_h[f] = (rng.random(len(state_h[f])) < 0.5).astype(np.uint8)
        rg.set_frames(group, [state_valid[f] for f in group], [state_h[f] for f in group])
        want, _ = O.graph_best_pair(n_frames, pairs, counts, offsets, q, t, state_valid, state_h, None)
        assert rg.query() == want == G.best_pair_for_valid(ctx, n_frames, pairs, counts, offsets, q, t, state_valid, state_h, None)
        rg.set_frames([], [], [])
        with pytest.raises(capi.EachamError):
            rg.set_frames([n_frames], [1], [np.zeros(3, np.uint8)])
        with pytest.raises(capi.EachamError):
            rg.set_frames([0], [1], [np.zeros(len(has3d[0]) + 1, np.uint8)])
        with pytest.raises(capi.EachamError):
            rg.set_frame(n_frames, True)
        with pytest.raises(capi.EachamError):
            rg.set_frame(0, True, np.zeros(len(has3d[0]) + 1, np.uint8))
        with pytest.raises(capi.EachamError):
            rg.query([n_frames])
    finally:
        rg.close()


def test_on_a_real_match_graph(ctx):
    """The matcher's own CSR output feeds the query without any re-packing."""
    sc = synth.make_scene(8, 2500, 4, seed=3)
    descs, ids = synth.make_frame_descriptors(sc, 400, 128, seed=3)
    for f, d in enumerate(descs):
        ctx.upload_descriptors(f, d)
    pairs = synth.all_pairs(8)
    counts, offsets, q, t, _ = ctx.match_all_pairs(pairs, min_dir=5, min_mutual=5)
    assert (counts > 0).sum() >= 4
    valid = np.array([1, 1, 1, 0, 0, 0, 0, 0], np.uint8)
    has3d = [(np.asarray(i) >= 0) & bool(valid[f]) for f, i in enumerate(ids)]   # keypoints that see a landmark
    got = G.best_pair_for_valid(ctx, 8, pairs, counts, offsets, q, t, valid, has3d)
    want, _ = O.graph_best_pair(8, pairs, counts, offsets, q, t, valid, has3d)
    assert got == want and got[2] > 0 and valid[got[0]] and not valid[got[1]]
    ctx.clear_descriptors()


def test_empty_and_invalid(ctx):
    z32, z64, zu = np.zeros(0, np.int32), np.zeros(1, np.int64), np.zeros(0, np.uint32)
    assert G.best_pair_for_valid(ctx, 3, np.zeros((0, 2), np.int32), z32, z64, zu, zu, np.ones(3, np.uint8),
                                 [np.zeros(2)] * 3) == (G.NONE, G.NONE, 0)
    pairs = np.array([[0, 1]], np.int32)
    with pytest.raises(capi.EachamError) as e:    # a match index beyond the frame's keypoints never reaches the kernel
        G.best_pair_for_valid(ctx, 2, pairs, np.array([1], np.int32), np.array([0, 1], np.int64), np.array([5], np.uint32),
                              np.array([0], np.uint32), np.array([1, 0], np.uint8), [np.zeros(2), np.zeros(2)])
    assert e.value.code == capi.ERR_INVALID
    with pytest.raises(capi.EachamError) as e:
        G.best_pair_for_valid(ctx, 2, np.array([[0, 7]], np.int32), np.array([0], np.int32), np.array([0, 0], np.int64), zu, zu,
                              np.array([1, 0], np.uint8), [np.zeros(2), np.zeros(2)])
    assert e.value.code == capi.ERR_INVALID
