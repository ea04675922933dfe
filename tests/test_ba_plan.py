"""CPU: the host-side analysis of the sparse reduced-system solve (eacham_amd/csrc/ba_plan.hpp) — ordering, panel
layout, symbolic factorisation, elimination tree, level schedule — compiled on its own with g++ and EXECUTED in
plain double arithmetic by tests/cpp/plan_driver.cpp on a random SPD matrix of the camera graph's pattern: the
schedule the device runs (leaf factors, level updates in item order, raw-tile back-substitution with the right-hand
side as row 63 of the root panel) must reproduce a dense Cholesky solve, for every ordering.
The counterpart in the reference: GTSAM's COLAMD ordering + multifrontal Cholesky, selected through
LevenbergMarquardtParams::SetCeresDefaults (modules/sfm/reconstruction/BundleAdjuster.cpp:182-190)."""
import json
import os
import subprocess

import numpy as np
import pytest

from eacham_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("plan") / "plan_driver")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "cpp", "plan_driver.cpp")],
                   check=True, capture_output=True)
    return exe


def run(exe, adj, ordering, seed=1):
    i, j = np.nonzero(np.triu(adj, 1))
    text = f"{adj.shape[0]} {len(i)} {ordering} {seed}\n" + "".join(f"{a} {b}\n" for a, b in zip(i, j))
    r = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-400:], r.stderr[-400:])
    return json.loads(r.stdout)


def band(n, w, second=None):
    a = np.zeros((n, n), bool)
    idx = np.arange(n)
    d = np.abs(idx[:, None] - idx[None, :])
    a[(d > 0) & (d <= w)] = True
    if second is not None:
        a[(d >= second[0]) & (d <= second[1])] = True
    return a


def scene_graph(n_cams, n_lm, k, seed=synth.MASTER_SEED):
    sc = synth.make_scene(n_cams, n_lm, k, seed=seed)
    cams = sc["obs_cam"].reshape(n_lm, -1).astype(np.int64)
    a = np.zeros((n_cams, n_cams), bool)
    for x in range(cams.shape[1]):
        for y in range(cams.shape[1]):
            a[cams[:, x], cams[:, y]] = True
    np.fill_diagonal(a, False)
    return a


@pytest.mark.parametrize("n_cams", [0, 1, 3, 9, 10, 11, 19, 20, 21, 22, 31, 32, 43, 64])
@pytest.mark.parametrize("ordering", [0, 1, 2, 3])
def test_small_windows_and_panel_edges(driver, n_cams, ordering):
    """n = 6 n_cams + 5 around the panel edges: K and the right-hand-side row share the last panel with cameras or get
    a panel of their own, cameras straddle two panels."""
    out = run(driver, band(n_cams, 3), ordering)
    assert out["bad"] == 0 and out["spd"] == 1 and out["rel_err"] < 1e-12
    assert out["npan"] >= (6 * n_cams + 5 + 1 + 63) // 64   # room for the right-hand-side row
    if ordering == 1:
        assert out["levels"] == out["npan"]                  # the caller's order of a band: a path


def test_disconnected_and_complete_graphs(driver):
    two = np.zeros((40, 40), bool)
    two[:20, :20] = band(20, 4)
    two[20:, 20:] = band(20, 4)
    full = ~np.eye(30, dtype=bool)
    for adj in (two, full, np.zeros((25, 25), bool)):
        for ordering in (0, 1, 2, 3):
            out = run(driver, adj, ordering)
            assert out["bad"] == 0 and out["spd"] == 1 and out["rel_err"] < 1e-12
    # independent components are factorised side by side: the tree is no path
    assert run(driver, two, 3)["levels"] < run(driver, two, 1)["levels"]


def test_a_long_sequence_gets_a_bushy_tree(driver):
    """A 300-frame sequence (band of +-6 frames): the caller's order is a path of 29 panels; nested dissection halves the
    height and more, and the cost model picks it."""
    adj = band(300, 6)
    nat, nd, auto = (run(driver, adj, o) for o in (1, 3, 0))
    for out in (nat, nd, auto):
        assert out["bad"] == 0 and out["rel_err"] < 1e-12
    assert nat["levels"] == nat["npan"] == 29
    assert nd["levels"] <= 12 and auto["ordering"] == 3 and auto["est_us"] < 0.6 * nat["est_us"]


def test_metric_scene_graphs(driver):
    """The camera graphs of the BASELINE scenes: S200 (200 cameras on two turns of a helix: a band of +-9 plus a wide
    band one turn away — few separators: 19 panels as a path, 12 levels dissected) and config 4 (500 cameras: 47 -> 16)."""
    s200 = scene_graph(200, 50_000, 10)
    nat, auto = run(driver, s200, 1), run(driver, s200, 0)
    assert nat["bad"] == auto["bad"] == 0 and max(nat["rel_err"], auto["rel_err"]) < 1e-11
    assert nat["levels"] == nat["npan"] == 19 and auto["levels"] <= 13 and auto["ordering"] == 3
    assert auto["tile_updates"] < nat["tile_updates"]
    c4 = scene_graph(500, 100_000, 10, seed=4)
    nat, auto = run(driver, c4, 1), run(driver, c4, 0)
    assert nat["bad"] == auto["bad"] == 0 and max(nat["rel_err"], auto["rel_err"]) < 1e-11
    assert nat["levels"] == 47 and auto["levels"] <= 20 and auto["tile_updates"] < 0.3 * nat["tile_updates"]
