"""GPU: the C++ adapters (include/eacham/*.hpp — the reference-shaped IFeatureMatcher / RefineBA on top
of the C-ABI) compiled with g++ and checked against the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from eacham_amd import ba, synth
import oracle_api as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _build(tmp, name="adapter_driver", glue=False):
    """glue: the RefineBA / TriangulateFrame leg goes through include/eacham/ReferenceGlue.hpp — the reference-typed entry
    points — on stand-ins of Graph / Node / Map / cv::Mat (tests/cpp/ref_standins.hpp) instead of hand-filled views."""
    exe = os.path.join(tmp, name + ("_glue" if glue else ""))
    lib = os.path.join(ROOT, "eacham_amd", "lib")
    cmd = ["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "cpp"),
           *(["-DEACHAM_TEST_GLUE"] if glue else []), os.path.join(ROOT, "tests", "cpp", name + ".cpp"),
           "-o", exe, "-L" + lib, "-leacham_hip", "-Wl,-rpath," + lib, "-lpthread"]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def _vec(f, dtype):
    n = struct.unpack("q", f.read(8))[0]
    return np.frombuffer(f.read(n * np.dtype(dtype).itemsize), dtype=dtype).copy()


def write_adapter_fixture(tmp):
    """Inputs of tests/cpp/adapter_driver.cpp (also used, with a stub of the C-ABI, by tests/test_sanitizers.py)."""
    import types
    # ---- matcher inputs ----
    sc = synth.make_scene(4, 300, 3, seed=17)
    descs, _ = synth.make_frame_descriptors(sc, 200, 128, seed=17)
    descs[2] = descs[2][:150]
    # ---- a graph/map with the cases RefineBA's walk distinguishes ----
    bsc = synth.make_scene(7, 400, 4, seed=23)
    ids = [10, 11, 12, 20, 21, 30, 31]                 # node ids are not dense
    valid = [1, 1, 1, 1, 0, 1, 1]                      # node 21 is not valid -> skipped as a neighbour
    fixed = [1, 0, 0, 0, 0, 0, 0]
    neighbours = {12: [10, 11, 20, 21, 30]}            # local window around node 12 (31 is not a neighbour)
    current = 12
    n_lm = 400
    status = np.ones(n_lm, bool); status[::9] = False  # not-yet-triangulated landmarks are filtered
    observers = bsc["observers"].copy(); observers[5::11] = 1   # < 2 observers -> filtered
    cam_of = {i: k for k, i in enumerate(ids)}
    kp = {i: [] for i in ids}; p3 = {i: [] for i in ids}
    for o in range(len(bsc["obs_cam"])):
        node = ids[int(bsc["obs_cam"][o])]
        p3[node].append((len(kp[node]), int(bsc["obs_lm"][o]) + 100))   # landmark ids offset by 100
        kp[node].append(bsc["obs_uv"][o].astype(np.float32))
    K9 = np.array([bsc["K"][0], 0, bsc["K"][2], 0, bsc["K"][1], bsc["K"][3], 0, 0, 1.0])

    fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("ii", len(descs), 128))
        for d in descs:
            f.write(struct.pack("i", d.shape[0])); f.write(np.ascontiguousarray(d, np.float32).tobytes())
        f.write(struct.pack("i", len(ids)))
        for k, i in enumerate(ids):
            f.write(struct.pack("Iii", i, valid[k], fixed[k])); f.write(bsc["T_init"][k].astype(np.float64).tobytes())
            f.write(struct.pack("i", len(kp[i]))); f.write(np.array(kp[i], np.float32).reshape(-1).tobytes())
            f.write(struct.pack("i", len(p3[i])))
            for a, b in p3[i]:
                f.write(struct.pack("II", a, b))
            nb = neighbours.get(i, [])
            f.write(struct.pack("i", len(nb))); f.write(np.array(nb, np.uint32).tobytes())
        f.write(struct.pack("i", n_lm))
        for j in range(n_lm):
            f.write(struct.pack("I", j + 100)); f.write(bsc["points_init"][j].astype(np.float64).tobytes())
            f.write(struct.pack("iI", int(status[j]), int(observers[j])))
        f.write(K9.tobytes()); f.write(struct.pack("i", current))
    return types.SimpleNamespace(**locals())


@pytest.mark.parametrize("glue", [False, True])
def test_cpp_adapters_against_oracle(tmp_path, glue):
    tmp = str(tmp_path)
    exe = _build(tmp, glue=glue)
    fx = write_adapter_fixture(tmp)
    (fin, fout, descs, bsc, ids, valid, fixed, neighbours, n_lm, status, observers, kp, p3) = (
        fx.fin, fx.fout, fx.descs, fx.bsc, fx.ids, fx.valid, fx.fixed, fx.neighbours, fx.n_lm, fx.status, fx.observers, fx.kp, fx.p3)
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr

    with open(fout, "rb") as f:
        for a, b in [(0, 1), (1, 0), (0, 2), (2, 1)]:
            got = _vec(f, np.uint32).reshape(-1, 2)
            q, t = O.match_directed(descs[a], descs[b])
            assert np.array_equal(got[:, 0], q) and np.array_equal(got[:, 1], t)
        counts, gq, gt = _vec(f, np.int32), _vec(f, np.uint32), _vec(f, np.uint32)
        want = O.match_all_pairs(descs, synth.all_pairs(4), min_dir=5, min_mutual=5)
        assert np.array_equal(counts, want[0]) and np.array_equal(gq, want[2]) and np.array_equal(gt, want[3]) and counts.sum() > 0
        bp = _vec(f, np.uint32)
        valid4 = np.array([1, 1, 0, 0], np.uint8)
        has3d = [(np.arange(d.shape[0]) % 3 == 0) & bool(valid4[k]) for k, d in enumerate(descs)]
        wbp, _ = O.graph_best_pair(4, synth.all_pairs(4), want[0], want[1], want[2], want[3], valid4, has3d)
        assert tuple(bp.tolist()) == wbp and wbp[2] > 0
        meta, K_out = _vec(f, np.float64), _vec(f, np.float64)
        Ts, Ps, st_out = _vec(f, np.float64).reshape(-1, 4, 4), _vec(f, np.float64).reshape(-1, 3), _vec(f, np.int32)

    # expected: the same window built here, independently, then solved by the oracle
    window = [12] + [i for i in neighbours[12] if valid[ids.index(i)]]
    lm_index, pts, obs_c, obs_p, uv, nobs = {}, [], [], [], [], []
    for w, node in enumerate(window):
        for a, lm in sorted(p3[node]):
            j = lm - 100
            if not status[j] or observers[j] < 2:
                continue
            if lm not in lm_index:
                lm_index[lm] = len(pts); pts.append(bsc["points_init"][j]); nobs.append(observers[j])
            obs_c.append(w); obs_p.append(lm_index[lm]); uv.append(kp[node][a].astype(np.float64))
    A = ba.BaArrays(np.array([bsc["T_init"][ids.index(i)] for i in window]), np.array([fixed[ids.index(i)] for i in window], np.int32),
                    np.array(pts), np.array(nobs, np.int32), np.array(obs_c, np.uint32), np.array(obs_p, np.uint32),
                    np.array(uv), bsc["K"])
    ref = O.ba_solve(A, ba.OptimizerConfig.refine_ba())
    assert meta[0] == 0 and int(meta[5]) == len(window) and int(meta[6]) == len(pts) and int(meta[7]) == len(obs_c)
    assert (int(meta[3]), int(meta[4])) == (ref.outer_iterations, ref.inner_iterations)
    assert np.isclose(meta[2], ref.final_error, rtol=1e-9)
    assert np.allclose([K_out[0], K_out[4], K_out[2], K_out[5]], ref.K, rtol=1e-9)
    for w, node in enumerate(window):                     # window poses updated ...
        assert np.abs(Ts[ids.index(node)] - ref.cam_T_wc[w]).max() < 1e-7
    for k, i in enumerate(ids):                           # ... everything else untouched
        if i not in window:
            assert np.array_equal(Ts[k], bsc["T_init"][k])
    for lm, idx in lm_index.items():
        assert np.abs(Ps[lm - 100] - ref.points[idx]).max() < 1e-7 and st_out[lm - 100] == 1
    untouched = [j for j in range(n_lm) if (j + 100) not in lm_index]
    assert np.array_equal(Ps[untouched], bsc["points_init"][untouched]) and np.array_equal(st_out[untouched], status[untouched].astype(np.int32))


def write_tri_fixture(tmp):
    """Inputs of tests/cpp/tri_driver.cpp (also used, with a stub of the C-ABI, by tests/test_sanitizers.py)."""
    import types
    sc = synth.make_scene(8, 300, 5, seed=31, pixel_noise=1.0)
    ids = [3, 4, 7, 9, 12, 15, 16, 20]
    valid = [1, 1, 1, 0, 1, 1, 1, 1]
    cur = 5                                            # node 15 is being inserted
    frame_id = ids[cur]
    K, T = sc["K"], sc["T_true"]
    K9 = np.array([K[0], 0, K[2], 0, K[1], K[3], 0, 0, 1.0])
    max_err, min_angle, min_obs = np.float32(4.0), np.float32(3.0 * 3.141592 / 180.0), 2
    # keypoints per node, in observation order; keypoint pixels are stored as float (cv::Point2f)
    kp = {i: [] for i in ids}
    kp_of = {}                                         # (node, landmark) -> keypoint index
    for o in range(len(sc["obs_cam"])):
        node = ids[int(sc["obs_cam"][o])]
        lm = int(sc["obs_lm"][o])
        kp_of[(node, lm)] = len(kp[node])
        uv = sc["obs_uv"][o].astype(np.float32)
        if lm % 7 == 0 and node == frame_id:
            uv = uv + np.float32(40.0)                  # bad keypoints in the new frame
        kp[node].append(uv)
    # existing map: landmarks 0,3,6,... were triangulated earlier from the other frames
    p3 = {i: {} for i in ids}
    mpts = {}
    for lm in range(0, 300, 3):
        obs = {n: kp_of[(n, lm)] for n in ids if n != frame_id and (n, lm) in kp_of}
        if len(obs) < 2:
            continue
        if lm % 2 == 0:
            obs = dict(list(obs.items())[:2])          # only two observers: fails the `> 2` gate
        pid = 1000 + lm
        mpts[pid] = {"p": sc["points_true"][lm] + 0.001, "valid": True, "obs": dict(obs)}
        for n, k in obs.items():
            p3[n][k] = pid
    factors = {}
    for n in ids:
        if n == frame_id:
            continue
        mm = [(kp_of[(frame_id, lm)], kp_of[(n, lm)]) for lm in range(300) if (frame_id, lm) in kp_of and (n, lm) in kp_of]
        if mm:
            factors[n] = mm

    fin, fout = os.path.join(tmp, "tin.bin"), os.path.join(tmp, "tout.bin")
    first = sorted(factors)[0]
    lm_single = [lm for lm in range(300) if sum((n, lm) in kp_of for n in ids) >= 3][0]
    single = [(n, kp_of[(n, lm_single)]) for n in ids if (n, lm_single) in kp_of]
    with open(fin, "wb") as f:
        f.write(struct.pack("i", len(ids)))
        for k, i in enumerate(ids):
            f.write(struct.pack("Ii", i, valid[k])); f.write(T[k].astype(np.float64).tobytes())
            f.write(struct.pack("i", len(kp[i]))); f.write(np.array(kp[i], np.float32).reshape(-1).tobytes())
            f.write(struct.pack("i", len(p3[i])))
            for a, b in sorted(p3[i].items()):
                f.write(struct.pack("II", a, b))
            fs = factors if i == frame_id else {}
            f.write(struct.pack("i", len(fs)))
            for other in sorted(fs):
                f.write(struct.pack("Ii", other, len(fs[other]))); f.write(np.array(fs[other], np.uint32).tobytes())
        f.write(struct.pack("i", len(mpts)))
        for pid in sorted(mpts):
            f.write(struct.pack("I", pid)); f.write(mpts[pid]["p"].astype(np.float64).tobytes())
            f.write(struct.pack("ii", 1, len(mpts[pid]["obs"])))
            for a, b in sorted(mpts[pid]["obs"].items()):
                f.write(struct.pack("II", a, b))
        f.write(K9.tobytes()); f.write(struct.pack("IIff", frame_id, min_obs, max_err, min_angle))
        f.write(struct.pack("i", len(single)))
        for n, k in single:
            f.write(T[ids.index(n)].astype(np.float64).tobytes()); f.write(kp[n][k].astype(np.float64).tobytes())
    return types.SimpleNamespace(**locals())


@pytest.mark.parametrize("glue", [False, True])
def test_cpp_triangulate_frame_against_oracle_walk(tmp_path, glue):
    """TriangulatorHip.hpp: TriangulatePointRansac + TriangulateFrame (Triangulator.cpp:96-300)."""
    tmp = str(tmp_path)
    exe = _build(tmp, "tri_driver", glue=glue)
    fx = write_tri_fixture(tmp)
    (fin, fout, sc, ids, valid, frame_id, K, T, max_err, min_angle, min_obs, kp, p3, mpts, factors, single) = (
        fx.fin, fx.fout, fx.sc, fx.ids, fx.valid, fx.frame_id, fx.K, fx.T, fx.max_err, fx.min_angle, fx.min_obs, fx.kp, fx.p3, fx.mpts,
        fx.factors, fx.single)
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr

    # ---- the same walk in Python on the oracle ----
    Tt = T.reshape(-1, 16)
    row = {n: k for k, n in enumerate(ids)}
    full = {}
    reobs = 0
    for other in sorted(factors):
        if not valid[row[other]]:
            continue
        for m1, m2 in factors[other]:
            if m2 in p3[other]:
                pid = p3[other][m2]
                err = O.reprojection_errors(Tt, [row[frame_id]], [mpts[pid]["p"]], [kp[frame_id][m1].astype(np.float64)], K)[0]
                if len(mpts[pid]["obs"]) > 2 and err < max_err:
                    p3[frame_id][m1] = pid
                    mpts[pid]["obs"][frame_id] = m1
                    reobs += 1
                    continue
            full.setdefault(m1, {})[frame_id] = m1
            full[m1][other] = m2
    tp, fr, uv, tracks = [0], [], [], []
    for m1 in sorted(full):
        if len(full[m1]) < min_obs:
            continue
        for n in sorted(full[m1]):
            fr.append(row[n]); uv.append(kp[n][full[m1][n]].astype(np.float64))
        tp.append(len(fr)); tracks.append(full[m1])
    pts, status, _ = O.tri_tracks(Tt, tp, fr, np.array(uv), K, max_err, min_angle)
    next_id = max(mpts)
    added = 0
    for t, obs in enumerate(tracks):
        if status[t] == 3:
            next_id += 1
            mpts[next_id] = {"p": pts[t], "valid": True, "obs": {}}
            for n in sorted(obs):
                k = obs[n]
                if k in p3[n]:
                    old = p3[n][k]
                    mpts[old]["obs"].pop(n, None); mpts[old]["valid"] = False
                p3[n][k] = next_id
                mpts[next_id]["obs"][n] = k
            added += 1
    assert added > 10 and reobs > 5 and added < len(tracks)

    with open(fout, "rb") as f:
        sgl, X = _vec(f, np.int32), _vec(f, np.float64)
        spts, sst, smask = O.tri_tracks(np.array([T[ids.index(n)].reshape(16) for n, _ in single]), [0, len(single)], np.arange(len(single)),
                                        np.array([kp[n][k].astype(np.float64) for n, k in single]), K, max_err, min_angle)
        assert sgl[0] == (sst[0] & 1) and np.allclose(X, spts[0], rtol=1e-9)
        assert sgl[1:].tolist() == (smask.tolist() if smask.any() else [])
        meta = _vec(f, np.uint32)
        assert meta.tolist() == [len(tracks), added, reobs, next_id]
        for n in ids:
            flat = _vec(f, np.uint32).reshape(-1, 2)
            assert dict(map(tuple, flat.tolist())) == p3[n], n
        gid, gvalid, gP, gobs = _vec(f, np.uint32), _vec(f, np.uint32), _vec(f, np.float64).reshape(-1, 3), _vec(f, np.uint32)
    assert gid.tolist() == sorted(mpts)
    pos = 0
    for j, pid in enumerate(gid.tolist()):
        assert bool(gvalid[j]) == mpts[pid]["valid"], pid
        assert np.allclose(gP[j], mpts[pid]["p"], rtol=1e-9, atol=1e-12), pid
        cnt = int(gobs[pos]); flat = gobs[pos + 1:pos + 1 + 2 * cnt].reshape(-1, 2); pos += 1 + 2 * cnt
        assert dict(map(tuple, flat.tolist())) == mpts[pid]["obs"], pid


def _write_frames(path, descs, dim):
    with open(path, "wb") as f:
        f.write(struct.pack("ii", len(descs), dim))
        for d in descs:
            f.write(struct.pack("i", d.shape[0])); f.write(np.ascontiguousarray(d, np.float32).tobytes())


@pytest.mark.parametrize("kind", ["u8", "f32"])
def test_drop_in_match_under_the_reference_call_pattern(tmp_path, kind):
    """apps/sfm/main.cpp:84-109 on the drop-in: one std::async(&Match) per ORDERED pair from a pool of threads on one
    shared instance. Every result must equal the oracle's directed match; every frame is uploaded exactly once (the
    cache by buffer address) and concurrent callers are served in batches (fewer launches than calls)."""
    tmp = str(tmp_path)
    exe = _build(tmp, "match_async_driver")
    sc = synth.make_scene(7, 500, 4, seed=29)
    if kind == "u8":
        descs, _ = synth.make_frame_descriptors(sc, 240, 128, seed=29)
        descs[3] = descs[3][:97]
        descs[5] = descs[5][:1]            # a single-row train frame: nothing passes the ratio test against it
    else:
        base = synth.unit_float_descriptors(240, 64, 1, 99)
        descs = [synth.unit_float_descriptors(240, 64, 1, f, shared=base[:150]) for f in range(7)]
        descs[3] = descs[3][:97]
    dim = descs[0].shape[1]
    fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
    _write_frames(fin, descs, dim)
    r = subprocess.run([exe, fin, fout, "8", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    F = len(descs)
    with open(fout, "rb") as f:
        for i in range(F):
            for j in range(F):
                if i == j:
                    continue
                got = _vec(f, np.uint32).reshape(-1, 2)
                qo, to = O.match_directed(descs[i], descs[j], force_f32=2 if kind == "f32" else 0)
                assert np.array_equal(got[:, 0], qo) and np.array_equal(got[:, 1], to), (i, j)
        seconds, calls, batches, uploads, hits = np.frombuffer(f.read(40), dtype=np.float64)
    assert calls == 2 * F * (F - 1)                    # two repeats
    assert uploads == (F if kind == "u8" else F + 1)   # f32: the first frame is tried on the int8 path once
    assert hits >= 2 * calls - uploads - 1 and batches < calls and seconds > 0
