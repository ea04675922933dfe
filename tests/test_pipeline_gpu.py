"""GPU: the stages composed the way eacham's incremental loop composes them (apps/sfm/main.cpp):
match all pairs -> match graph -> next-pair query -> tracks -> triangulation -> bundle adjustment,
every stage through the C-ABI, every stage checked against its oracle, the result against ground truth."""
import numpy as np
import pytest

import oracle_api as O
from eacham_amd import HipContext, ba, synth
from eacham_amd import graph as G
from eacham_amd import triangulate as tri

pytestmark = pytest.mark.gpu


def _tracks_from_matches(n_frames, kpts, pairs, counts, offsets, q, t):
    """Union-find over (frame, keypoint) nodes; keeps tracks that touch a frame at most once."""
    parent = np.arange(n_frames * kpts)

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for p, (f1, f2) in enumerate(pairs.tolist()):
        for k in range(int(offsets[p]), int(offsets[p]) + int(counts[p])):
            a, b = find(f1 * kpts + int(q[k])), find(f2 * kpts + int(t[k]))
            if a != b:
                parent[max(a, b)] = min(a, b)
    groups = {}
    for node in range(n_frames * kpts):
        r = find(node)
        groups.setdefault(r, []).append(node)
    tracks = []
    for nodes in groups.values():
        frames = [n // kpts for n in nodes]
        if len(nodes) >= 2 and len(set(frames)) == len(frames):
            tracks.append(sorted(nodes))
    return sorted(tracks)


def test_match_graph_tracks_triangulation_ba():
    n_frames, kpts, dim = 10, 400, 128
    sc = synth.make_scene(n_frames, 1200, 5, seed=11, pixel_noise=0.7)
    descs, ids = synth.make_frame_descriptors(sc, kpts, dim, seed=11)
    # pixel of every keypoint: the scene's observation of (frame, landmark); distractors get a random pixel
    uv_of = {(int(c), int(l)): sc["obs_uv"][o] for o, (c, l) in enumerate(zip(sc["obs_cam"], sc["obs_lm"]))}
    rnd = synth.rng_uniform(11, 900, (n_frames, kpts, 2)) * 800.0
    kp_uv = np.array([[uv_of.get((f, int(ids[f][k])), rnd[f, k]) for k in range(kpts)] for f in range(n_frames)])
    pairs = synth.all_pairs(n_frames)

    with HipContext(0) as ctx:
        # 1. match graph
        for f, d in enumerate(descs):
            ctx.upload_descriptors(f, d)
        counts, offsets, q, t, stats = ctx.match_all_pairs(pairs, min_dir=5, min_mutual=5)
        want = O.match_all_pairs(descs, pairs, min_dir=5, min_mutual=5)
        for g, w in zip((counts, offsets, q, t, stats), want[:5]):
            assert np.array_equal(g, w)
        assert (counts > 0).sum() >= 15
        # every mutual match joins two keypoints of the same landmark (the descriptors are that clean)
        for p, (f1, f2) in enumerate(pairs.tolist()):
            sl = slice(int(offsets[p]), int(offsets[p]) + int(counts[p]))
            if counts[p]:
                assert (np.asarray(ids[f1])[q[sl]] == np.asarray(ids[f2])[t[sl]]).mean() > 0.99

        # 2. next pair to add, frames 0..2 already reconstructed
        valid = np.zeros(n_frames, np.uint8)
        valid[:3] = 1
        has3d = [(np.asarray(ids[f]) >= 0) & bool(valid[f]) for f in range(n_frames)]
        bp = G.best_pair_for_valid(ctx, n_frames, pairs, counts, offsets, q, t, valid, has3d)
        assert bp == O.graph_best_pair(n_frames, pairs, counts, offsets, q, t, valid, has3d)[0]
        assert valid[bp[0]] and not valid[bp[1]] and bp[2] > 20

        # 3. tracks -> triangulation with the (known) poses
        tracks = _tracks_from_matches(n_frames, kpts, pairs, counts, offsets, q, t)
        assert len(tracks) > 500
        track_ptr = np.zeros(len(tracks) + 1, np.int32)
        track_ptr[1:] = np.cumsum([len(tr) for tr in tracks])
        nodes = np.concatenate(tracks)
        obs_frame, obs_kp = (nodes // kpts).astype(np.uint32), nodes % kpts
        obs_uv = kp_uv[obs_frame, obs_kp]
        T = sc["T_true"].reshape(-1, 16)
        args = (T, track_ptr, obs_frame, obs_uv, sc["K"], 4.0, 3.0 * 3.141592 / 180.0)
        pts, status, masks = tri.triangulate_tracks(ctx, *args)
        opts, ostatus, omasks = O.tri_tracks(*args)
        assert np.array_equal(status, ostatus) and np.array_equal(masks, omasks)
        ok = (status & 2) != 0   # every observation an inlier (the world-z > 0 quirk of bit 0 is irrelevant to geometry)
        assert ok.sum() > 300
        lm_of_track = np.array([ids[tr[0] // kpts][tr[0] % kpts] for tr in tracks])
        good = ok & (lm_of_track >= 0)
        err_tri = np.linalg.norm(pts[good] - sc["points_true"][lm_of_track[good]], axis=1)
        assert np.median(err_tri) < 0.01

        # 4. bundle adjustment of the triangulated structure from perturbed poses
        sel = np.nonzero(good)[0]
        pid = -np.ones(len(tracks), np.int64)
        pid[sel] = np.arange(sel.size)
        track_of_obs = np.repeat(np.arange(len(tracks)), np.diff(track_ptr))
        keep = good[track_of_obs]
        arrays = ba.BaArrays(sc["T_init"], sc["fixed"], pts[sel], np.diff(track_ptr)[sel].astype(np.int32),
                             obs_frame[keep].astype(np.uint32), pid[track_of_obs[keep]].astype(np.uint32), obs_uv[keep], sc["K"])
        cfg = ba.OptimizerConfig.refine_ba()
        out = ba.RefineBA(ctx, arrays, cfg)
        ref = O.ba_solve(arrays, cfg)
    assert (out.outer_iterations, out.inner_iterations) == (ref.outer_iterations, ref.inner_iterations)
    assert np.abs(out.points - ref.points).max() < 1e-5 * np.abs(ref.points).max()
    assert np.abs(out.cam_T_wc - ref.cam_T_wc).max() < 1e-5
    assert out.final_error < 0.05 * out.initial_error
    # geometry: camera centres return to the truth (gauge held by the fixed first camera and the priors)
    centre = lambda Tm: -np.einsum("nji,nj->ni", Tm[:, :3, :3], Tm[:, :3, 3])
    c_true, c_init, c_out = centre(sc["T_true"]), centre(sc["T_init"]), centre(out.cam_T_wc.reshape(-1, 4, 4))
    assert np.linalg.norm(c_out - c_true, axis=1).mean() < 0.5 * np.linalg.norm(c_init - c_true, axis=1).mean()
