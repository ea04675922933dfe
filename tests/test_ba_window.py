"""CPU: the structure of the dense form of the Schur stage for a local window (eacham_amd/csrc/ba_window.hpp; the per-frame RefineBA
of apps/sfm/main.cpp:207 -> modules/sfm/reconstruction/BundleAdjuster.cpp:123-145) — compiled on its own with g++ and walked by
tests/cpp/window_driver.cpp against the header's definition; the bounds that make eacham_ba_prepare fall back to the pair lists
(group sizes that do not fit a CU's LDS or the form's bound on its partials, more than 24 cameras, a camera that sees a landmark
twice). The arithmetic that runs on this structure is held against the oracle on the GPU
(tests/test_ba_gpu.py::test_the_dense_form_of_a_local_window_solves_the_same_system)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("window") / "window_driver")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "cpp", "window_driver.cpp")],
                   check=True, capture_output=True)
    return exe


def run(exe, nc, nl, seed, max_obs, dup=0, rows=0):
    r = subprocess.run([exe], input=f"{nc} {nl} {seed} {max_obs} {dup} {rows}\n", capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-400:], r.stderr[-400:])
    return json.loads(r.stdout)


@pytest.mark.parametrize("nc,nl,max_obs,rows", [(1, 20, 1, 0), (3, 40, 3, 0), (19, 2000, 8, 0), (19, 2000, 8, 96), (19, 2000, 8, 200),
                                                (24, 1500, 24, 0), (12, 300, 12, 64)])
def test_the_structure_is_the_one_the_header_defines(driver, nc, nl, max_obs, rows):
    for seed in (1, 2):
        out = run(driver, nc, nl, seed, max_obs, 0, rows)
        assert out["built"] and out["bad"] == 0, out
        assert out["rows"] == (rows or out["default_rows"]) and out["rows"] % 4 == 0
        assert out["lds_bytes"] <= 160 * 1024 and out["partial_bytes"] <= 12 << 20


def test_what_the_form_does_not_cover_is_refused(driver):
    """eacham_ba_prepare then keeps the pair lists."""
    assert run(driver, 25, 500, 1, 8)["built"] is False                 # more cameras than a partial has blocks for
    assert run(driver, 19, 2000, 1, 8, dup=1)["built"] is False         # a camera sees a landmark twice
    assert run(driver, 19, 2000, 1, 8, rows=256)["built"] is False      # 256 rows of a 19-camera window: 167 KB of LDS
    assert run(driver, 19, 3000, 1, 8, rows=64)["built"] is False       # ~240 groups of 64 rows: more than 12 MB of partials
    assert run(driver, 19, 2000, 1, 8, rows=62)["built"] is False       # not a multiple of four / below the smallest group
    assert run(driver, 5, 30, 1, 0)["built"] is False                   # no observation at all
