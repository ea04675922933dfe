"""GPU: the HIP matcher (through the C-ABI) against the CPU oracle — indices must be bit-exact."""
import os

import numpy as np
import pytest

from eacham_amd import EachamError, FeatureMatcherHip, synth
from eacham_amd import capi
import np_reference as R
import oracle_api as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "match_golden.npz")


def _upload(ctx, descs):
    ctx.clear_descriptors()
    for f, d in enumerate(descs):
        ctx.upload_descriptors(f, d)


def _assert_csr_equal(got, want):
    names = ["counts", "offsets", "q", "t", "stats"]
    for n, g, w in zip(names, got[:5], want[:5]):
        assert np.array_equal(g, w), f"{n} differs"


def test_golden_fixture(hip_ctx):
    g = np.load(GOLD)
    descs = [g[f"desc{f}"].astype(np.float32) for f in range(4)]
    _upload(hip_ctx, descs)
    for tag, (md, mm) in {"ref": (30, 30), "low": (5, 5)}.items():
        c, o, q, t, st = hip_ctx.match_all_pairs(g["pairs"], float(g["ratio"]), md, mm)
        assert np.array_equal(c, g[f"counts_{tag}"]) and np.array_equal(o, g[f"offsets_{tag}"])
        assert np.array_equal(q, g[f"q_{tag}"]) and np.array_equal(t, g[f"t_{tag}"])
        assert np.array_equal(st, g[f"stats_{tag}"])
    for a, b in [(0, 1), (1, 0), (2, 3), (3, 2)]:
        q, t = hip_ctx.match_pair(a, b)
        assert np.array_equal(q, g[f"dir_{a}_{b}_q"]) and np.array_equal(t, g[f"dir_{a}_{b}_t"])


@pytest.mark.parametrize("n1,n2,dim", [(1, 2, 16), (31, 33, 64), (64, 64, 128), (200, 150, 128),
                                       (97, 203, 256), (513, 700, 256), (1100, 520, 128), (300, 280, 48)])
def test_directed_parity(hip_ctx, n1, n2, dim):
    A = synth.random_u8_descriptors(n1, dim, 21, 0)
    B = synth.random_u8_descriptors(n2, dim, 21, 1)
    m = min(n1, n2) // 2
    B[:m] = np.clip(A[:m] + np.rint(6 * synth.rng_normal(5, 3, (m, dim))), 0, 255)
    _upload(hip_ctx, [A, B])
    for a, b, X, Y in [(0, 1, A, B), (1, 0, B, A)]:
        q, t = hip_ctx.match_pair(a, b)
        qo, to = O.match_directed(X, Y)
        assert np.array_equal(q, qo) and np.array_equal(t, to)
    assert len(O.match_directed(A, B)[0]) >= m // 2 or m < 4


def test_extreme_values_and_ties(hip_ctx):
    """0/255 saturated rows exercise the full 25-bit rank range; duplicates exercise tie-breaks."""
    dim = 256
    A = synth.random_u8_descriptors(130, dim, 8, 0)
    B = synth.random_u8_descriptors(190, dim, 8, 1)
    A[0], A[1], A[2] = 0, 255, 0
    A[2, ::2] = 255
    B[0], B[1], B[3] = 255, 0, 255
    B[3, ::2] = 0
    B[100:110] = A[20:30]          # exact duplicates (distance 0 -> 0/0 never passes)
    B[110:120] = A[20:30]          # and a second copy: tie between 100.. and 110..
    B[120:125] = np.clip(A[40:45] + 1, 0, 255)
    B[125:130] = np.clip(A[40:45] + 1, 0, 255)  # equal non-zero distances: lower index must win
    _upload(hip_ctx, [A, B])
    for a, b, X, Y in [(0, 1, A, B), (1, 0, B, A)]:
        q, t = hip_ctx.match_pair(a, b)
        qo, to = O.match_directed(X, Y)
        qr, tr = R.directed(X, Y)
        assert np.array_equal(q, qo) and np.array_equal(t, to)
        assert np.array_equal(q, qr) and np.array_equal(t, tr)
    c, o, q, t, st = hip_ctx.match_all_pairs(np.array([[0, 1], [1, 0]]), min_dir=1, min_mutual=0)
    want = O.match_all_pairs([A, B], np.array([[0, 1], [1, 0]]), min_dir=1, min_mutual=0)
    _assert_csr_equal((c, o, q, t, st), want)


def test_ragged_and_empty_frames(hip_ctx):
    sc = synth.make_scene(6, 500, 4, seed=31)
    descs, _ = synth.make_frame_descriptors(sc, 300, 128, seed=31)
    descs[1] = descs[1][:37]
    descs[2] = descs[2][:0]            # empty frame
    descs[3] = descs[3][:1]            # single row: nothing can pass the ratio test against it
    descs[4] = descs[4][:64]
    pairs = np.concatenate([synth.all_pairs(6), np.array([[5, 0], [4, 1]], dtype=np.int32)])
    _upload(hip_ctx, descs)
    got = hip_ctx.match_all_pairs(pairs, min_dir=3, min_mutual=2)
    want = O.match_all_pairs(descs, pairs, min_dir=3, min_mutual=2)
    _assert_csr_equal(got, want)
    assert got[0].sum() > 0
    assert len(hip_ctx.match_pair(0, 2)[0]) == 0 and len(hip_ctx.match_pair(2, 0)[0]) == 0
    assert len(hip_ctx.match_pair(0, 3)[0]) == 0


def test_scene_parity_reference_thresholds(hip_ctx):
    """config[0]-like: 10 frames, SIFT-sized 128-D, literal thresholds 0.8 / 30 / 30."""
    sc = synth.make_scene(10, 3000, 5, seed=synth.MASTER_SEED)
    descs, _ = synth.make_frame_descriptors(sc, 1500, 128)
    pairs = synth.all_pairs(10)
    _upload(hip_ctx, descs)
    got = hip_ctx.match_all_pairs(pairs)
    want = O.match_all_pairs(descs, pairs)
    _assert_csr_equal(got, want)
    assert (got[0] > 30).sum() >= 5


def test_frames_beyond_one_column_chunk(hip_ctx):
    """> 4096 rows per frame (config[0]: up to 15000 SIFT features): the sweep is split into column
    chunks of 4096 and the per-chunk row results are merged; ties across the chunk border included."""
    A = synth.random_u8_descriptors(5000, 64, 41, 0)
    B = synth.random_u8_descriptors(4500, 64, 41, 1)
    m = 2000
    B[:m] = np.clip(A[:m] + np.rint(5 * synth.rng_normal(9, 3, (m, 64))), 0, 255)
    B[4200:4300] = B[100:200]            # duplicates on both sides of the 4096 border: lower index wins
    A[4097] = A[3]
    _upload(hip_ctx, [A, B])
    for a, b, X, Y in [(0, 1, A, B), (1, 0, B, A)]:
        q, t = hip_ctx.match_pair(a, b)
        qo, to = O.match_directed(X, Y)
        assert np.array_equal(q, qo) and np.array_equal(t, to) and len(q) > 1000
    got = hip_ctx.match_all_pairs(np.array([[0, 1]]))
    want = O.match_all_pairs([A, B], np.array([[0, 1]]))
    _assert_csr_equal(got, want)
    hip_ctx.clear_descriptors()
    with pytest.raises(EachamError) as e:
        hip_ctx.upload_descriptors(0, np.zeros((16385, 16), np.float32))
    assert e.value.code == capi.ERR_UNSUPPORTED
    hip_ctx.clear_descriptors()


def test_sift_feature_cap_of_config1(hip_ctx):
    """configs[0]: config/SfmConfigNerf.json:10 caps SIFT at 15 000 features per frame. Two frames at the cap, 128-D
    integers: four column chunks of 4096, 59 row blocks of 256 per direction, through eacham_match_pair both ways and
    eacham_match_all_pairs, bit-exact against the oracle (planted correspondences, duplicates across chunk borders)."""
    n, m = 15000, 6000
    A = synth.random_u8_descriptors(n, 128, 71, 0)
    B = synth.random_u8_descriptors(n, 128, 71, 1)
    perm = synth.rng_permutation(71, 5, n)[:m]
    B[perm] = np.clip(A[:m] + np.rint(6 * synth.rng_normal(71, 3, (m, 128))), 0, 255)
    B[12300:12310] = B[4090:4100]        # duplicates in chunks 0/1 and 3: the lower index wins the tie
    A[14999] = A[8191]
    _upload(hip_ctx, [A, B])
    for a, b, X, Y in [(0, 1, A, B), (1, 0, B, A)]:
        q, t = hip_ctx.match_pair(a, b)
        qo, to = O.match_directed(X, Y)
        assert np.array_equal(q, qo) and np.array_equal(t, to) and len(q) > 4000
    got = hip_ctx.match_all_pairs(np.array([[0, 1]]))
    want = O.match_all_pairs([A, B], np.array([[0, 1]]))
    _assert_csr_equal(got, want)
    assert got[0][0] > 4000
    hip_ctx.clear_descriptors()


def _with_norm_parity(D, parity):
    """Forces the parity of every row's centred squared norm (= the parity of its count of odd
    values, 128 being even): parity 0/1 per row, or None to leave the row alone."""
    D = D.copy()
    odd = (D.astype(np.int64) % 2).sum(1) % 2
    for r in range(D.shape[0]):
        if parity[r] is not None and odd[r] != parity[r]:
            D[r, 0] += 1 if D[r, 0] < 255 else -1
    return D


@pytest.mark.parametrize("n1,n2,mode", [(300, 280, "even"), (300, 280, "odd"), (256, 512, "split8"),
                                        (257, 255, "split8"), (290, 301, "one_odd"), (33, 65, "mixed"),
                                        (1, 3, "mixed"), (600, 520, "mixed"), (96, 96, "even_then_odd")])
def test_parity_sorted_layout_edges(hip_ctx, n1, n2, mode):
    """Frames are stored sorted by the parity of the squared norm: all-even and all-odd
    frames, class sizes on and off the 32 / 256-row boundaries, a single odd row, tiny frames; exact
    duplicates across parity classes cannot exist, within a class they must keep the lower index."""
    dim = 128
    A = synth.random_u8_descriptors(n1, dim, 77, 0)
    B = synth.random_u8_descriptors(n2, dim, 77, 1)
    m = min(n1, n2) // 2
    B[:m] = np.clip(A[:m] + np.rint(5 * synth.rng_normal(6, 3, (m, dim))), 0, 255)
    if n2 > 40:
        B[30:34] = B[10:14]  # duplicated train rows (same parity by construction): ties to the lower index
    par = {"even": lambda n: [0] * n, "odd": lambda n: [1] * n, "one_odd": lambda n: [0] * (n - 1) + [1],
           "split8": lambda n: [0] * 256 + [1] * (n - 256) if n > 256 else [0] * n,
           "even_then_odd": lambda n: [0] * (n // 2) + [1] * (n - n // 2), "mixed": lambda n: [None] * n}[mode]
    A, B = _with_norm_parity(A, par(n1)), _with_norm_parity(B, par(n2))
    if n2 > 40:
        B[30:34] = B[10:14]
    _upload(hip_ctx, [A, B])
    for a, b, X, Y in [(0, 1, A, B), (1, 0, B, A)]:
        q, t = hip_ctx.match_pair(a, b)
        qo, to = O.match_directed(X, Y)
        assert np.array_equal(q, qo) and np.array_equal(t, to)
    for md in (0, 5):
        got = hip_ctx.match_all_pairs(np.array([[0, 1], [1, 0]]), min_dir=md, min_mutual=md)
        want = O.match_all_pairs([A, B], np.array([[0, 1], [1, 0]]), min_dir=md, min_mutual=md)
        _assert_csr_equal(got, want)
    hip_ctx.clear_descriptors()


@pytest.mark.parametrize("dim", [64, 128, 256])
def test_row_sweep_every_tile_count_and_parity_boundary(hip_ctx, dim):
    """The row sweep (match_sweep_kernel) is a software pipeline unrolled six tiles deep with a separate last
    call and a once-per-sweep parity boundary: train frames of 1 .. 15 tiles (every residue of the unrolling, one tile short
    and one over a tile edge), with the even / odd boundary at the first tile, at the last, in the middle and absent, all
    frames against all frames in ONE call (ragged sizes in one batch), both forms of the column direction (conftest)."""
    sizes = [1, 31, 32, 33, 64, 95, 97, 128, 160, 190, 224, 257, 288, 320, 350, 384, 417, 448, 480]
    rng_rows = synth.random_u8_descriptors(max(sizes), dim, 123, 0)
    descs = []
    for k, n in enumerate(sizes):
        D = np.clip(rng_rows[:n] + np.rint(6 * synth.rng_normal(123, 10 + k, (n, dim))), 0, 255).astype(np.float32)
        mode = k % 4   # 0: parities as they fall, 1: all even (no boundary), 2: all odd (boundary at tile 0), 3: one odd row (boundary at the last tile)
        par = [None] * n if mode == 0 else [0] * n if mode == 1 else [1] * n if mode == 2 else [0] * (n - 1) + [1]
        D = _with_norm_parity(D, par)
        if n > 40:
            D[n - 3] = D[5]   # a duplicate far apart (same parity: identical rows): the lower index must win
        descs.append(D)
    _upload(hip_ctx, descs)
    pairs = np.array([[a, b] for a in range(len(sizes)) for b in range(len(sizes)) if a != b], dtype=np.int32)
    for md, mm in ((1, 0), (3, 2), (30, 30)):
        got = hip_ctx.match_all_pairs(pairs, min_dir=md, min_mutual=mm)
        want = O.match_all_pairs(descs, pairs, min_dir=md, min_mutual=mm)
        _assert_csr_equal(got, want)
    assert got[0].sum() > 0
    hip_ctx.clear_descriptors()


def _ctx_with_env(**env):
    """A context of its own created under the given environment switches (they are read once, at eacham_ctx_create)."""
    from eacham_amd import HipContext
    old = {k: os.environ.get(k) for k in env}
    try:
        os.environ.update(env)
        return HipContext(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("form", ["exact", "bound"])
@pytest.mark.parametrize("dim", [64, 128, 256])
def test_both_forms_of_the_row_sweep_at_every_dimension(form, dim):
    """The lean form's row sweep either keeps every row's exact top-2 or — its bound form, the default up to 128-D — sixteen partial
    minima per row whose second smallest bounds the row's runner-up from above; the rows that do not fail the ratio test against
    the bound get their exact {minimum, tile, runner-up} from a pass of their own (match_rowpick_kernel +
    match_colverify_kernel<KS, true>). Both forms forced at 64 / 128 / 256-D (EACHAM_MATCH_SWEEP_FORM) against the oracle on: the
    ragged tile-count frames with every position of the parity boundary; train frames whose only two rows are ADJACENT (one subset
    of the bound form: the bound is a padding value, the row must go to the exact pass) with and without a passing ratio; a
    duplicated train row (the runner-up equals the minimum: ratio 1) and a row whose runner-up sits in the minimum's own subset."""
    ctx = _ctx_with_env(EACHAM_MATCH_SWEEP_FORM=form)
    try:
        sizes = [1, 2, 33, 64, 97, 160, 257, 350, 480]
        base = synth.random_u8_descriptors(max(sizes) + 8, dim, 321, 0)
        descs = []
        for k, n in enumerate(sizes):
            D = np.clip(base[:n] + np.rint(6 * synth.rng_normal(321, 20 + k, (n, dim))), 0, 255).astype(np.float32)
            mode = k % 4
            par = [None] * n if mode == 0 else [0] * n if mode == 1 else [1] * n if mode == 2 else [0] * (n - 1) + [1]
            descs.append(_with_norm_parity(D, par))
        # two-row train frames: the rows are neighbours in the stored order, one subset of the bound form
        near = np.clip(base[5:6] + np.rint(2 * synth.rng_normal(321, 90, (1, dim))), 0, 255).astype(np.float32)
        far = np.clip(255 - base[5:6], 0, 255).astype(np.float32)
        descs.append(_with_norm_parity(np.vstack([near, far]), [0, 0]))                  # row 5 of the others matches `near`: passes
        descs.append(_with_norm_parity(np.vstack([near, near.copy()]), [0, 0]))         # a duplicate: runner-up == minimum, ratio 1
        # the runner-up in the minimum's own subset: rows 0..3 of a tile go to one lane's accumulators 0..3 (one group of four)
        D = descs[4].copy()
        D[1] = np.clip(D[0] + np.rint(1.5 * synth.rng_normal(321, 91, (dim,))), 0, 255)
        descs.append(D)
        _upload(ctx, descs)
        nf = len(descs)
        pairs = np.array([[a, b] for a in range(nf) for b in range(nf) if a != b], dtype=np.int32)
        for md, mm in ((1, 0), (2, 1), (30, 30)):
            got = ctx.match_all_pairs(pairs, min_dir=md, min_mutual=mm)
            want = O.match_all_pairs(descs, pairs, min_dir=md, min_mutual=mm)
            _assert_csr_equal(got, want)
        assert got[0].sum() > 0
        for a, b in ((0, 9), (4, 9), (5, 10), (11, 4), (4, 11), (9, 3)):                  # directed lists through the same sweep
            q, t = ctx.match_pair(a, b)
            wq, wt = O.match_directed(descs[a], descs[b])
            assert np.array_equal(q, wq) and np.array_equal(t, wt), (a, b)
    finally:
        ctx.close()


def test_ratio_range(hip_ctx):
    """The mutual entry points take 0 < ratio <= 1; the directed one takes any ratio
    and must agree with the oracle also when ties pass (ratio > 1: the lower index wins)."""
    A = synth.random_u8_descriptors(120, 64, 5, 0)
    B = synth.random_u8_descriptors(140, 64, 5, 1)
    B[50:60] = B[20:30]
    B[:40] = np.clip(A[:40] + np.rint(4 * synth.rng_normal(2, 3, (40, 64))), 0, 255)
    _upload(hip_ctx, [A, B])
    with pytest.raises(EachamError) as e:
        hip_ctx.match_all_pairs(np.array([[0, 1]]), ratio=1.5)
    assert e.value.code == capi.ERR_INVALID
    for ratio in (1.0, 0.95, 0.6):
        got = hip_ctx.match_all_pairs(np.array([[0, 1]]), ratio=ratio, min_dir=0, min_mutual=0)
        want = O.match_all_pairs([A, B], np.array([[0, 1]]), ratio=ratio, min_dir=0, min_mutual=0)
        _assert_csr_equal(got, want)
    for ratio in (1.5, 1.0):
        q, t = hip_ctx.match_pair(1, 0, ratio)
        qo, to = O.match_directed(B, A, ratio)
        assert np.array_equal(q, qo) and np.array_equal(t, to)
    hip_ctx.clear_descriptors()


def test_device_side_pair_list_with_a_bad_frame_id(hip_ctx):
    """eacham_match_all_pairs_dev cannot validate a pair list that lives on the device: a pair naming a frame
    that is not resident must yield no match, never fault, and surface at the next eacham_ctx_sync."""
    import torch
    A = synth.random_u8_descriptors(300, 64, 31, 0)
    B = synth.random_u8_descriptors(280, 64, 31, 1)
    B[:150] = np.clip(A[:150] + np.rint(4 * synth.rng_normal(3, 3, (150, 64))), 0, 255)
    _upload(hip_ctx, [A, B])
    dev = torch.device("cuda", 0)
    pairs = np.array([[0, 1], [0, 7], [1, 0], [-3, 1], [5000000, 0]], np.int32)
    ext = torch.cuda.ExternalStream(hip_ctx.stream, device=dev)
    with torch.cuda.stream(ext):
        pd = torch.from_numpy(pairs).to(dev)
        counts = torch.full((len(pairs),), -1, dtype=torch.int32, device=dev)
        offsets = torch.zeros(len(pairs) + 1, dtype=torch.int64, device=dev)
        total = torch.zeros(1, dtype=torch.int64, device=dev)
        edges = torch.zeros(2 * 1000, dtype=torch.int32, device=dev)
        hip_ctx.match_all_pairs_dev(pd.data_ptr(), len(pairs), counts.data_ptr(), offsets.data_ptr(), edges.data_ptr(), 1000,
                                    total.data_ptr(), min_dir=5, min_mutual=5)
        with pytest.raises(EachamError) as e:
            hip_ctx.sync()
        assert e.value.code == capi.ERR_INVALID
        hip_ctx.sync()  # the error is reported once
    want = O.match_all_pairs([A, B], np.array([[0, 1], [1, 0]]), min_dir=5, min_mutual=5)
    c = counts.cpu().numpy()
    assert c.tolist() == [int(want[0][0]), 0, int(want[0][1]), 0, 0] and want[0][0] > 50
    assert int(total.item()) == int(want[0].sum())
    hip_ctx.clear_descriptors()


def test_feature_matcher_interface(hip_ctx):
    """FeatureMatcherFlann-shaped adapter: Match(d1, d2) -> {query: train}."""
    A = synth.random_u8_descriptors(150, 128, 3, 0)
    B = np.clip(A[::-1] + np.rint(4 * synth.rng_normal(5, 4, A.shape)), 0, 255).astype(np.float32)
    m = FeatureMatcherHip(0.8, context=hip_ctx)
    got = m.Match(A, B)
    q, t = O.match_directed(A, B)
    assert got == dict(zip(q.tolist(), t.tolist())) and len(got) > 100
    # float descriptors go through the same interface (fp32 path)
    base = synth.unit_float_descriptors(120, 256, 2, 9)
    Af = synth.unit_float_descriptors(120, 256, 2, 0, shared=base)
    Bf = synth.unit_float_descriptors(140, 256, 2, 1, shared=base)
    got = m.Match(Af, Bf)
    q, t = O.match_directed(Af, Bf, force_f32=2)
    assert got == dict(zip(q.tolist(), t.tolist())) and len(got) > 60
    hip_ctx.clear_descriptors()


def test_errors(hip_ctx):
    hip_ctx.clear_descriptors()
    with pytest.raises(EachamError) as e:
        hip_ctx.upload_descriptors(0, np.full((4, 32), 0.5, np.float32))
    assert e.value.code == capi.ERR_NOT_INTEGER
    with pytest.raises(EachamError) as e:
        hip_ctx.upload_descriptors(0, np.zeros((4, 24), np.float32))
    assert e.value.code == capi.ERR_UNSUPPORTED
    hip_ctx.clear_descriptors()
    hip_ctx.upload_descriptors(0, np.zeros((4, 32), np.float32))
    with pytest.raises(EachamError) as e:
        hip_ctx.match_pair(0, 7)
    assert e.value.code == capi.ERR_INVALID
    hip_ctx.clear_descriptors()


def test_full_size_properties(hip_ctx):
    """BASELINE config[1] frame size (2000 x 256): properties that need no oracle at this size,
    plus an oracle check on one pair."""
    sc = synth.make_scene(6, 6000, 4, seed=77)
    descs, ids = synth.make_frame_descriptors(sc, 2000, 256, seed=77)
    pairs = synth.all_pairs(6)
    _upload(hip_ctx, descs)
    c, o, q, t, st = hip_ctx.match_all_pairs(pairs)
    # symmetry: matching (j,i) yields the inverse edge list (Connect(n2,n1,best21), main.cpp:145)
    c2, o2, q2, t2, st2 = hip_ctx.match_all_pairs(pairs[:, ::-1].copy())
    assert np.array_equal(c, c2)
    for p in range(len(pairs)):
        a = sorted(zip(q[o[p]:o[p + 1]].tolist(), t[o[p]:o[p + 1]].tolist()))
        b = sorted(zip(t2[o2[p]:o2[p + 1]].tolist(), q2[o2[p]:o2[p + 1]].tolist()))
        assert a == b
        # sortedness and injectivity
        qq = q[o[p]:o[p + 1]]
        assert np.all(np.diff(qq.astype(np.int64)) > 0) and len(set(t[o[p]:o[p + 1]].tolist())) == len(qq)
        # every mutual match joins two observations of the same landmark in this synthetic scene
        i, j = pairs[p]
        la, lb = ids[i][qq], ids[j][t[o[p]:o[p + 1]]]
        assert np.array_equal(la, lb) and np.all(la >= 0)
    assert c.sum() > 1000
    # self-match: identical frames -> every distance-0 best is rejected by the ratio test (0/x or 0/0)
    hip_ctx.upload_descriptors(6, descs[0])
    qs, ts = hip_ctx.match_pair(0, 6)
    assert np.array_equal(qs, ts)  # each row's nearest is its copy; d0 = 0 passes 0/d1 < 0.8
    qo, to = O.match_directed(descs[0], descs[0])
    assert np.array_equal(qs, qo) and np.array_equal(ts, to)
    want = O.match_all_pairs(descs, pairs[:2])
    got = hip_ctx.match_all_pairs(pairs[:2])
    _assert_csr_equal(got, want)
    hip_ctx.clear_descriptors()


# ---- float descriptors (SuperPoint / LightGlue style): fp32 MFMA path ---------------------------------
def _float_frames(ns, dim, seed=61):
    base = synth.unit_float_descriptors(max(ns), dim, seed, 99)
    return [synth.unit_float_descriptors(n, dim, seed, k, shared=base[: n // 2]) for k, n in enumerate(ns)]


def _upload_f32(ctx, descs):
    ctx.clear_descriptors()
    for f, d in enumerate(descs):
        ctx.upload_descriptors_f32(f, d)


@pytest.mark.parametrize("ns,dim", [((300, 257), 256), ((130, 64, 1, 0, 97), 128), ((200, 333), 60), ((2000, 1900), 256)])
def test_f32_path_is_bit_exact_against_the_dot_form_oracle(hip_ctx, ns, dim):
    descs = _float_frames(ns, dim)
    _upload_f32(hip_ctx, descs)
    for a in range(len(ns)):
        for b in range(len(ns)):
            if a == b or (len(ns) > 2 and abs(a - b) > 1):
                continue
            q, t = hip_ctx.match_pair(a, b)
            qo, to = O.match_directed(descs[a], descs[b], force_f32=2)
            assert np.array_equal(q, qo) and np.array_equal(t, to), (a, b)
    pairs = synth.all_pairs(len(ns))
    got = hip_ctx.match_all_pairs(pairs, min_dir=3, min_mutual=2)
    want = O.match_all_pairs(descs, pairs, min_dir=3, min_mutual=2, force_f32=2)
    _assert_csr_equal(got, want)
    assert got[0].sum() > 0 or min(ns) < 8
    hip_ctx.clear_descriptors()


def test_f32_pair_beyond_8192_rows(hip_ctx):
    """The float path at the upper end of its range (limit 16 384 rows): an 8 200-row frame against a 3 000-row one,
    both directions and the mutual form, bit-exact against the dot-form oracle."""
    descs = _float_frames((8200, 3000), 256, seed=17)
    _upload_f32(hip_ctx, descs)
    for a, b in ((0, 1), (1, 0)):
        q, t = hip_ctx.match_pair(a, b)
        qo, to = O.match_directed(descs[a], descs[b], force_f32=2)
        assert np.array_equal(q, qo) and np.array_equal(t, to), (a, b)
        assert len(q) > 500
    got = hip_ctx.match_all_pairs(np.array([[0, 1]]))
    want = O.match_all_pairs(descs, np.array([[0, 1]]), force_f32=2)
    _assert_csr_equal(got, want)
    hip_ctx.clear_descriptors()


def test_f32_path_agrees_with_sum_of_squared_differences(hip_ctx):
    """The dot-product form rounds differently from OpenCV's sum (a-b)^2; on well-separated
    descriptors the match lists must coincide (index-agreement report of SURVEY.md §8(d), mode F32)."""
    descs = _float_frames((1500, 1400), 256, seed=5)
    _upload_f32(hip_ctx, descs)
    q, t = hip_ctx.match_pair(0, 1)
    qs, ts = O.match_directed(descs[0], descs[1], force_f32=1)
    got, ref = dict(zip(q.tolist(), t.tolist())), dict(zip(qs.tolist(), ts.tolist()))
    same = sum(1 for k, v in ref.items() if got.get(k) == v)
    agreement = same / max(len(ref), 1)
    print(f"f32 index agreement with the SSD form: {same}/{len(ref)} = {agreement:.5f}; extra {len(got) - same}")
    assert len(ref) > 500 and agreement >= 0.999 and abs(len(got) - len(ref)) <= max(2, len(ref) // 500)
    hip_ctx.clear_descriptors()


def test_f32_ties_and_kind_mixing(hip_ctx):
    A = synth.unit_float_descriptors(70, 64, 3, 0)
    B = np.concatenate([synth.unit_float_descriptors(40, 64, 3, 1), A[:8], A[:8]])  # duplicated train rows
    _upload_f32(hip_ctx, [A, B])
    for a, b, X, Y in [(0, 1, A, B), (1, 0, B, A)]:
        q, t = hip_ctx.match_pair(a, b)
        qo, to = O.match_directed(X, Y, force_f32=2)
        assert np.array_equal(q, qo) and np.array_equal(t, to)
    with pytest.raises(EachamError) as e:   # int8 and fp32 frames cannot be resident together
        hip_ctx.upload_descriptors(2, np.zeros((4, 64), np.float32))
    assert e.value.code == capi.ERR_UNSUPPORTED
    hip_ctx.clear_descriptors()
    hip_ctx.upload_descriptors(0, np.zeros((4, 64), np.float32))
    with pytest.raises(EachamError):
        hip_ctx.upload_descriptors_f32(1, A)
    hip_ctx.clear_descriptors()


def test_batched_directed_matches_equal_single_calls(hip_ctx):
    """eacham_match_pairs_directed: many ordered pairs in one launch sequence = that many FeatureMatcherFlann::Match
    calls (what the C++ adapter funnels concurrent callers into). Ragged, empty and single-row frames included."""
    sc = synth.make_scene(6, 500, 4, seed=31)
    descs, _ = synth.make_frame_descriptors(sc, 300, 128, seed=31)
    descs[1] = descs[1][:37]
    descs[2] = descs[2][:0]
    descs[3] = descs[3][:1]
    pairs = [(i, j) for i in range(6) for j in range(6) if i != j] + [(0, 0)]
    got = hip_ctx.match_pairs_directed(descs, pairs)
    for (i, j), m in zip(pairs, got):
        qo, to = O.match_directed(descs[i], descs[j])
        assert m == dict(zip(qo.tolist(), to.tolist())), (i, j)
        q1, t1 = hip_ctx.match_pair(i, j)
        assert np.array_equal(q1, qo) and np.array_equal(t1, to)
    assert sum(len(m) for m in got) > 200
    assert got[-1] == {q: q for q in range(300)}                 # a frame against itself: d0 = 0 < 0.8 d1 always passes
    assert hip_ctx.match_pairs_directed(descs, []) == []


def test_documented_limits_are_errors_not_wrong_answers(hip_ctx):
    """D <= 256, one descriptor kind and one dimension class resident at a time, <= 16384 rows."""
    hip_ctx.clear_descriptors()
    for bad_dim in (272, 512):
        with pytest.raises(EachamError) as e:
            hip_ctx.upload_descriptors(0, np.zeros((8, bad_dim), np.float32))
        assert e.value.code == capi.ERR_UNSUPPORTED
        with pytest.raises(EachamError) as e:
            hip_ctx.upload_descriptors_f32(0, np.zeros((8, bad_dim), np.float32))
        assert e.value.code == capi.ERR_UNSUPPORTED
    hip_ctx.upload_descriptors(0, synth.random_u8_descriptors(40, 128, 1, 0))
    with pytest.raises(EachamError) as e:   # a 256-D frame beside a 128-D one: another k-step class
        hip_ctx.upload_descriptors(1, synth.random_u8_descriptors(40, 256, 1, 1))
    assert e.value.code == capi.ERR_UNSUPPORTED
    hip_ctx.upload_descriptors(1, synth.random_u8_descriptors(40, 112, 1, 1))   # 112 and 128 share the class (zero padding)
    q, t = hip_ctx.match_pair(0, 0)
    assert np.array_equal(q, np.arange(40)) and np.array_equal(t, np.arange(40))
    with pytest.raises(EachamError) as e:
        hip_ctx.upload_descriptors_f32(2, np.zeros((16385, 32), np.float32))
    assert e.value.code == capi.ERR_UNSUPPORTED
    hip_ctx.clear_descriptors()
