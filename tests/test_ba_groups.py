"""CPU: the landmark-major structure of the Schur stage (eacham_amd/csrc/ba_groups.hpp) — landmark order, groups of bounded rows,
block-sorted padded entry chunks, segment masks, per-block partial lists — compiled on its own with g++ and EXECUTED in plain
doubles by tests/cpp/groups_driver.cpp: what ba_schur_groups + ba_assemble_groups compute from it must equal the direct sum
S_ab = sum_j Et_a(j) Et_b(j)^T over every pair of rows of every landmark (the landmark elimination behind
LevenbergMarquardtOptimizer::optimize, modules/sfm/reconstruction/BundleAdjuster.cpp:182-216), incl. the calibration border and
the K corner as blocks of the pseudo-camera."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("groups") / "groups_driver")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "cpp", "groups_driver.cpp")],
                   check=True, capture_output=True)
    return exe


def run(exe, nc, nl, seed, max_obs, dup):
    r = subprocess.run([exe], input=f"{nc} {nl} {seed} {max_obs} {dup}\n", capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-400:], r.stderr[-400:])
    return json.loads(r.stdout)


@pytest.mark.parametrize("nc,nl,max_obs,dup", [(1, 5, 1, 0), (7, 60, 6, 0), (20, 1500, 10, 0), (20, 1500, 10, 1), (200, 9000, 12, 0),
                                               (40, 3000, 30, 1), (3, 400, 3, 1)])
def test_structure_reproduces_the_direct_schur_sums(driver, nc, nl, max_obs, dup):
    for seed in (1, 2):
        out = run(driver, nc, nl, seed, max_obs, dup)
        assert out["built"] and out["bad"] == 0, out
        assert out["worst"] < 1e-12, out
        assert out["max_rows"] <= out["rows"] and out["max_lm"] <= out["rows"] // 4
        assert out["padded"] >= out["entries"] and out["blocks"] >= 2 * nc + 1


def test_large_problems_take_480_row_groups(driver):
    out = run(driver, 100, 6000, 3, 10, 0)      # > 32 768 rows
    assert out["built"] and out["rows"] == 480 and out["bad"] == 0 and out["worst"] < 1e-12


def test_a_landmark_too_heavy_for_a_group_is_refused(driver):
    out = run(driver, 200, 40, 5, 150, 0)       # up to 150 observers: more entries than a 128-row group may hold
    assert out["built"] is False                # eacham_ba_prepare then keeps the pair lists of rounds 1-4
