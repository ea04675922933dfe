"""CPU: the matching oracle against an independent numpy statement and the golden fixture."""
import os

import numpy as np
import pytest

from eacham_amd import synth
import np_reference as R
import oracle_api as O

GOLD = os.path.join(os.path.dirname(__file__), "golden", "match_golden.npz")


@pytest.mark.parametrize("n1,n2,dim", [(1, 2, 16), (33, 65, 64), (200, 150, 128), (97, 203, 256)])
def test_directed_matches_numpy(n1, n2, dim):
    A = synth.random_u8_descriptors(n1, dim, 7, 0)
    B = synth.random_u8_descriptors(n2, dim, 7, 1)
    B[: min(n1, n2) // 2] = np.clip(A[: min(n1, n2) // 2] + synth.rng_normal(3, 9, (min(n1, n2) // 2, dim)).round() * 4, 0, 255)
    q, t = O.match_directed(A, B)
    qr, tr = R.directed(A, B)
    assert np.array_equal(q, qr) and np.array_equal(t, tr)
    assert len(q) > 0 or n1 < 4


def test_integer_and_float_paths_agree():
    A = synth.random_u8_descriptors(120, 128, 11, 0)
    B = synth.random_u8_descriptors(140, 128, 11, 1)
    B[:60] = np.clip(A[:60] + 3, 0, 255)
    a = O.match_mutual(A, B, min_dir=1, min_mutual=0)
    b = O.match_mutual(A, B, min_dir=1, min_mutual=0, force_f32=True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_ties_resolve_to_lower_train_index():
    A = synth.random_u8_descriptors(40, 64, 5, 0)
    B = np.concatenate([synth.random_u8_descriptors(50, 64, 5, 1), A[:10], A[:10]])  # duplicated train rows
    idx, d0, d1 = O.knn2(A, B)
    assert np.array_equal(idx[:10], np.arange(50, 60))  # first copy wins
    assert np.all(d0[:10] == 0) and np.all(d1[:10] == 0)
    q, t = O.match_directed(A, B)  # 0/0 = NaN never passes the ratio test
    assert not np.any(q < 10)


def test_edge_cases_empty_and_short_train():
    A = synth.random_u8_descriptors(10, 32, 1, 0)
    empty = np.zeros((0, 32), np.float32)
    assert len(O.match_directed(A, empty)[0]) == 0
    assert len(O.match_directed(empty, A)[0]) == 0
    assert len(O.match_directed(A, A[:1])[0]) == 0  # the reference would read m[1] out of bounds
    q, t, st = O.match_mutual(A, empty)
    assert len(q) == 0 and st.tolist() == [0, 0, 0, 0]


def test_mutual_thresholds_follow_main_cpp():
    sc = synth.make_scene(3, 120, 3, seed=99)
    descs, _ = synth.make_frame_descriptors(sc, 80, 64, seed=99)
    q, t, st = O.match_mutual(descs[0], descs[1])
    qr, tr, sr = R.mutual(descs[0], descs[1])
    assert np.array_equal(st, sr) and np.array_equal(q, qr) and np.array_equal(t, tr)
    n = int(st[2])
    # `> min_mutual` is strict (main.cpp:142), `>= min_dir` per direction (main.cpp:111)
    assert len(O.match_mutual(descs[0], descs[1], min_dir=1, min_mutual=n)[0]) == 0
    assert len(O.match_mutual(descs[0], descs[1], min_dir=1, min_mutual=n - 1)[0]) == n
    assert len(O.match_mutual(descs[0], descs[1], min_dir=int(st[0]), min_mutual=0)[0]) == n
    assert len(O.match_mutual(descs[0], descs[1], min_dir=int(max(st[0], st[1])) + 1, min_mutual=0)[0]) == 0


def test_golden_fixture():
    g = np.load(GOLD)
    descs = [g[f"desc{f}"].astype(np.float32) for f in range(4)]
    for tag, (md, mm) in {"ref": (30, 30), "low": (5, 5)}.items():
        c, o, q, t, st, _ = O.match_all_pairs(descs, g["pairs"], float(g["ratio"]), md, mm)
        assert np.array_equal(c, g[f"counts_{tag}"]) and np.array_equal(o, g[f"offsets_{tag}"])
        assert np.array_equal(q, g[f"q_{tag}"]) and np.array_equal(t, g[f"t_{tag}"])
        assert np.array_equal(st, g[f"stats_{tag}"])
    for a, b in [(0, 1), (1, 0), (2, 3), (3, 2)]:
        q, t = O.match_directed(descs[a], descs[b])
        assert np.array_equal(q, g[f"dir_{a}_{b}_q"]) and np.array_equal(t, g[f"dir_{a}_{b}_t"])
        qr, tr = R.directed(descs[a], descs[b])  # and the numpy statement agrees with the fixture
        assert np.array_equal(qr, g[f"dir_{a}_{b}_q"]) and np.array_equal(tr, g[f"dir_{a}_{b}_t"])


def test_threads_do_not_change_results():
    sc = synth.make_scene(5, 200, 3, seed=4)
    descs, _ = synth.make_frame_descriptors(sc, 64, 32, seed=4)
    pairs = synth.all_pairs(5)
    r1 = O.match_all_pairs(descs, pairs, min_dir=2, min_mutual=2, nthreads=1)
    r4 = O.match_all_pairs(descs, pairs, min_dir=2, min_mutual=2, nthreads=4)
    for x, y in zip(r1[:5], r4[:5]):
        assert np.array_equal(x, y)
