"""CPU: include/eacham/CvSampling.hpp — the sample stream of OpenCV's robust estimators (cv::RNG, getSubset, the homography's
checkSubset), which the E / H / PnP loops of TwoViewHip.hpp / PnPHip.hpp draw from by default. OpenCV is not in the image:
the recurrences are restated from memory of the 4.5.5 sources (unverifiable here; parity unpinned). What CAN be pinned is that
the header computes the recurrence it states: its first draws against values written down from an independent statement
(python integers), uniform() and getSubset against a literal python replay of the loops, checkSubset against numpy."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COEFF = 4164903690  # CV_RNG_COEFF

# the first 20 values of cv::RNG((uint64)-1).next(): state <- (uint32)state * 4164903690 + (state >> 32), output (uint32)state
FIRST_DRAWS = [130063605, 3133359004, 2578348940, 925327173, 1080261831, 2946015512, 94037301, 2298661280, 300167573, 43921110,
               776868985, 1162377994, 3771123546, 691074649, 1024279418, 1989440103, 882328646, 2642864120, 3691087718, 3930549720]


class PyRNG:
    def __init__(self, state=0xFFFFFFFFFFFFFFFF):
        self.state = state

    def next(self):
        self.state = ((self.state & 0xFFFFFFFF) * COEFF + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return self.state & 0xFFFFFFFF

    def uniform(self, a, b):
        return a if a == b else self.next() % (b - a) + a


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cvs") / "cvsampling_driver")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "cvsampling_driver.cpp"), "-o", out], check=True, capture_output=True)
    return out


def run(exe, *args, stdin=None):
    return subprocess.run([exe, *map(str, args)], input=stdin, capture_output=True, text=True, check=True).stdout.split("\n")[:-1]


def test_first_draws_of_the_generator(exe):
    rng = PyRNG()
    assert [rng.next() for _ in range(20)] == FIRST_DRAWS            # the written-down values are the recurrence's
    assert [int(x) for x in run(exe, "draws")] == FIRST_DRAWS        # and the header computes them


def test_uniform_and_get_subset_replay(exe):
    for n in (5, 292, 2000, 15000):
        rng = PyRNG()
        assert [int(x) for x in run(exe, "uniform", n, 50)] == [rng.uniform(0, n) for _ in range(50)]
    for n, m in ((2000, 5), (292, 4), (6, 5), (5, 5)):               # (small n: repeated indices are redrawn, as getSubset does)
        rng = PyRNG()
        want = []
        for _ in range(40):
            sub = []
            while len(sub) < m:
                v = rng.uniform(0, n)
                while v in sub:
                    v = rng.uniform(0, n)
                sub.append(v)
            want.append(sub)
        got = [[int(x) for x in line.split()] for line in run(exe, "subsets", n, m, 40)]
        assert got == want and all(len(set(s)) == m for s in got)


def test_homography_check_subset(exe):
    rng = np.random.default_rng(3)
    cases, want = [], []

    def det(p, t):
        return np.linalg.det(np.array([[p[t[0]][0], p[t[0]][1], 1.0], [p[t[1]][0], p[t[1]][1], 1.0], [p[t[2]][0], p[t[2]][1], 1.0]]))

    def collinear_last(p):
        i = 3
        for j in range(i):
            d1 = np.float32(p[j]) - np.float32(p[i])
            for k in range(j):
                d2 = np.float32(p[k]) - np.float32(p[i])
                dx1, dy1, dx2, dy2 = float(d1[0]), float(d1[1]), float(d2[0]), float(d2[1])
                if abs(dx2 * dy1 - dy2 * dx1) <= np.finfo(np.float32).eps * (abs(dx1) + abs(dy1) + abs(dx2) + abs(dy2)):
                    return True
        return False
    for c in range(200):
        src = rng.uniform(0, 800, (4, 2)).astype(np.float32)
        if c % 4 == 0:                                               # a similarity: orientation kept
            a = rng.uniform(0, 6.28)
            R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
            dst = (src @ R.T * 1.3 + 5).astype(np.float32)
        elif c % 4 == 1:                                             # a mirror of ONE point pair: mixed signs
            dst = src.copy(); dst[[0, 1]] = dst[[1, 0]]
        elif c % 4 == 2:                                             # the last point on the line through two earlier ones
            dst = rng.uniform(0, 800, (4, 2)).astype(np.float32)
            src[3] = src[0] + np.float32(0.5) * (src[1] - src[0])
        else:
            dst = rng.uniform(0, 800, (4, 2)).astype(np.float32)
        tt = [(0, 1, 2), (1, 2, 3), (0, 2, 3), (0, 1, 3)]
        neg = sum(det(src.astype(np.float64), t) * det(dst.astype(np.float64), t) < 0 for t in tt)
        ok = not collinear_last(src) and not collinear_last(dst) and neg in (0, 4)
        cases.append(" ".join(repr(float(v)) for v in np.concatenate([src.ravel(), dst.ravel()])))
        want.append(int(ok))
    got = [int(x) for x in run(exe, "check", stdin="\n".join(cases) + "\n")]
    assert got == want and 20 < sum(want) < 180                      # both verdicts occur
