"""include/eacham/TwoViewHip.hpp — the OpenCV calls of RecoverPoseTwoView (ReconstructionManager.cpp:47-183) on top of the C-ABI:
FindEssentialMat / FindHomography (LMedS over eacham_solve_minimal + eacham_score_hypotheses), DecomposeHomographyMat,
DecomposeEssentialMat / RecoverPose. CPU: the two decompositions (host-only math) against their defining equations and the
ground truth; GPU: the whole pipeline against the ground truth of a synthetic pair (OpenCV's own sampling is tied to its RNG:
parity unpinned, so the yardstick is the truth, not OpenCV's numbers)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from eacham_amd import synth
import score_cases as SC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def rot(w):
    return synth.so3_exp(np.asarray(w, float))


def test_decompositions_satisfy_their_equations_and_contain_the_truth(tmp_path):
    exe = str(tmp_path / "twoview_driver")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(CPP, "twoview_driver.cpp"), os.path.join(CPP, "stub_abi.cpp"), "-o", exe, "-lpthread"], check=True, capture_output=True)
    rng = np.random.default_rng(3)
    K = np.array([[960, 0, 400], [0, 960, 400], [0, 0, 1.0]])
    lines, truth = [], []
    for k in range(12):
        R = rot(rng.normal(size=3) * 0.3)
        t = rng.normal(size=3) * 0.5
        n = rng.normal(size=3) + np.array([0, 0, 3.0])
        n /= np.linalg.norm(n)
        d = 4.0
        Hn = R + np.outer(t, n) / d
        H = K @ Hn @ np.linalg.inv(K)
        H /= H[2, 2] * (1 if k % 2 == 0 else -1)            # H is only known up to scale AND sign
        lines.append("H " + " ".join(f"{x:.17g}" for x in H.ravel()) + " " + " ".join(f"{x:.17g}" for x in K.ravel()))
        E = SC.skew(t) @ R
        lines.append("E " + " ".join(f"{x:.17g}" for x in (E / np.linalg.norm(E)).ravel()))
        truth.append((R, t / d, n, Hn, t / np.linalg.norm(t)))
    lines.append("H " + " ".join(f"{x:.17g}" for x in (K @ rot([0.1, -0.2, 0.05]) @ np.linalg.inv(K)).ravel()) + " " + " ".join(f"{x:.17g}" for x in K.ravel()))
    rvecs = [np.array([0.3, -0.2, 0.9]), np.zeros(3), np.array([1e-11, 0, 0]), np.pi * np.array([0.6, -0.8, 0.0]), np.pi * np.array([0, 0, 1.0]),
             (np.pi - 1e-7) * np.array([1.0, 2.0, -2.0]) / 3.0]
    for rv in rvecs:                                                                        # cv::Rodrigues(R) of PnPHip.hpp
        lines.append("R " + " ".join(f"{x:.17g}" for x in rot(rv).ravel()))
    out = subprocess.run([exe, "decompose"], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout.split("\n")
    i = k = 0
    for k in range(12):
        assert out[i].split() == ["H", "4"]
        R, t, n, Hn, tdir = truth[k]
        best = 1e9
        for j in range(4):
            s = np.array(out[i + 1 + j].split(), float)
            Rs, ts, ns = s[:9].reshape(3, 3), s[9:12], s[12:15]
            assert np.abs(Rs @ Rs.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(Rs) - 1) < 1e-12 and abs(np.linalg.norm(ns) - 1) < 1e-12
            assert np.abs(Rs + np.outer(ts, ns) - Hn).max() < 1e-10                     # H_normalised = R + t n^T, exactly
            best = min(best, np.abs(Rs - R).max() + min(np.abs(ts - t).max() + np.abs(ns - n).max(), np.abs(ts + t).max() + np.abs(ns + n).max()))
        assert best < 1e-9                                                                # the motion that generated H is one of them
        i += 5
        assert out[i] == "E"
        v = np.array(out[i + 1].split(), float)
        R1, R2, td = v[:9].reshape(3, 3), v[9:18].reshape(3, 3), v[18:]
        assert min(np.abs(R1 - R).max(), np.abs(R2 - R).max()) < 1e-12 and min(np.abs(td - tdir).max(), np.abs(td + tdir).max()) < 1e-12
        assert abs(np.linalg.det(R1) - 1) < 1e-12 and abs(np.linalg.det(R2) - 1) < 1e-12
        i += 2
    assert out[i].split() == ["H", "1"]                                                   # a pure rotation: one solution, t = 0
    s = np.array(out[i + 1].split(), float)
    assert np.abs(s[:9].reshape(3, 3) - rot([0.1, -0.2, 0.05])).max() < 1e-12 and not s[9:12].any()
    i += 2
    for rv in rvecs:
        got = np.array(out[i].split()[1:], float)
        assert np.abs(rot(got) - rot(rv)).max() < 1e-6 and abs(np.linalg.norm(got) - np.linalg.norm(rv)) < 1e-6   # (the axis of a half turn has no sign)
        i += 1


def _vec(f, dtype):
    n = struct.unpack("q", f.read(8))[0]
    return np.frombuffer(f.read(n * np.dtype(dtype).itemsize), dtype=dtype).copy()


@pytest.mark.gpu
def test_two_view_pipeline_recovers_the_relative_pose(tmp_path):
    """RecoverPoseTwoView's OpenCV calls end to end on the device library: a general scene (E branch: findEssentialMat
    1000 LMedS iterations -> recoverPose) and a planar one (H branch: findHomography 100 iterations -> decomposeHomographyMat
    -> the triangulation vote of :100-150), 25 % gross outliers each."""
    exe = str(tmp_path / "twoview_driver")
    lib = os.path.join(ROOT, "eacham_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(CPP, "twoview_driver.cpp"), "-o", exe,
                    "-L" + lib, "-leacham_hip", "-Wl,-rpath," + lib, "-lpthread"], check=True, capture_output=True)
    cases = [SC.two_view_case(n=800, seed=31, outliers=0.25), SC.two_view_case(n=800, seed=32, outliers=0.25, planar=True, facing=True),
             # noise-free pixels (only their float rounding is left): the median is ~0 and the inlier rule rests on OpenCV's
             # `sigma = MAX(sigma, 0.001)` — without it the masks would shrink to the below-median half of the good matches
             SC.two_view_case(n=800, seed=33, outliers=0.25, noise=0.0), SC.two_view_case(n=800, seed=34, outliers=0.25, planar=True, facing=True, noise=0.0)]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        for c in cases:
            K = c["K"]
            f.write(struct.pack("i", 800)); f.write(c["uv1"].tobytes()); f.write(c["uv2"].tobytes())
            f.write(np.array([K[0], 0, K[2], 0, K[1], K[3], 0, 0, 1.0]).tobytes())
    r = subprocess.run([exe, "pipeline", fin, fout], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    with open(fout, "rb") as f:
        for k, c in enumerate(cases):
            meta, E, H = _vec(f, np.float64), _vec(f, np.float64).reshape(3, 3), _vec(f, np.float64).reshape(3, 3)
            emask, hmask = _vec(f, np.uint8), _vec(f, np.uint8)
            pose, hb = _vec(f, np.float64), _vec(f, np.float64)
            good = ~c["bad"]
            T21 = c["T21"]
            Rt, tt = T21[:3, :3], T21[:3, 3] / np.linalg.norm(T21[:3, 3])
            assert meta[0] == 1 and meta[3] == 1
            assert meta[6] == 89 and meta[7] == 72          # LMedS' fixed budgets: 1000 asked at 0.99 / 5 points, 100 at 0.999 / 4 points
            if k == 2:                                                       # noise-free, general: EVERY good match is an inlier of E
                assert emask[good].all() and int(emask.sum()) == int(good.sum()) + int(emask[c["bad"]].sum()) and emask[c["bad"]].mean() < 0.05
                continue
            if k == 3:                                                       # noise-free, planar: every good match is an inlier of H
                assert hmask[good].all() and hmask[c["bad"]].mean() < 0.05
                continue
            if k == 0:                                                       # general scene: the essential matrix and its pose
                Et = c["E"][0].reshape(3, 3) / np.linalg.norm(c["E"][0])
                assert min(np.abs(E - Et).max(), np.abs(E + Et).max()) < 0.02
                assert emask[good].mean() > 0.8 and emask[c["bad"]].mean() < 0.2
                R, t, ngood = pose[:9].reshape(3, 3), pose[9:12], pose[12]
                # (the best MINIMAL model, unrefined, as cv::findEssentialMat returns it: ~1 degree at 0.5 px noise)
                assert np.abs(R - Rt).max() < 0.03 and np.abs(t - tt).max() < 0.08 and ngood > 0.7 * good.sum()
            else:                                                            # planar scene: the homography and its decomposition
                Ht = c["H"][0].reshape(3, 3)
                p = np.c_[c["uv1"], np.ones(800)] @ H.T
                q = np.c_[c["uv1"], np.ones(800)] @ Ht.T
                assert np.median(np.linalg.norm(p[good, :2] / p[good, 2:] - q[good, :2] / q[good, 2:], axis=1)) < 2.0
                assert hmask[good].mean() > 0.8 and hmask[c["bad"]].mean() < 0.2
                nsol, best = int(hb[0]), int(hb[1])
                counts = hb[2:2 + nsol]
                assert nsol == 4 and best == int(np.argmax(counts)) and counts[best] > 0.5 * good.sum()   # first strict maximum (:139-144)
                sols = hb[2 + nsol:].reshape(nsol, 12)
                # the reference's vote only looks at camera 1 (z > 0 and the reprojection there, :113-127), so it cannot tell the
                # true motion from its mirror twins: what is checked is that the truth is one of the four and is not out-voted
                err = [np.abs(s_[:9].reshape(3, 3) - Rt).max() + np.abs(s_[9:] / np.linalg.norm(s_[9:]) - tt).max() for s_ in sols]
                near = int(np.argmin(err))
                assert err[near] < 0.2 and counts[near] > 0.5 * good.sum()


@pytest.mark.gpu
def test_solve_pnp_ransac_recovers_the_camera_pose(tmp_path):
    """SolvePnPRansac (PnPHip.hpp) = cv::solvePnPRansac(..., 10000, 4.0f, 0.999f, inliers, SOLVEPNP_EPNP) of RecoverPosePnP
    (ReconstructionManager.cpp:227-238): 30 % gross outliers, 0.8 px noise."""
    exe = str(tmp_path / "twoview_driver")
    lib = os.path.join(ROOT, "eacham_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(CPP, "twoview_driver.cpp"), "-o", exe,
                    "-L" + lib, "-leacham_hip", "-Wl,-rpath," + lib, "-lpthread"], check=True, capture_output=True)
    c = SC.pnp_case(n=500, seed=11, outliers=0.3)
    X, uv, K, T, bad = c["X"][1:], c["uv"][1:], c["K"], c["models"][0], c["bad"][1:]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("i", len(X))); f.write(np.ascontiguousarray(X).tobytes()); f.write(np.ascontiguousarray(uv).tobytes())
        f.write(np.array([K[0], 0, K[2], 0, K[1], K[3], 0, 0, 1.0]).tobytes())
    r = subprocess.run([exe, "pnp", fin, fout], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    with open(fout, "rb") as f:
        pose, inliers = _vec(f, np.float64), _vec(f, np.int32)
    ok, iters, R, rvec, t = pose[0], pose[1], pose[2:11].reshape(3, 3), pose[11:14], pose[14:17]
    assert ok == 1 and 5 <= iters < 200                       # 70 % inliers: OpenCV's adaptive budget stops after a few dozen samples
    assert np.abs(R - T[:9].reshape(3, 3)).max() < 2e-3 and np.abs(t - T[9:]).max() < 5e-3
    assert np.abs(rot(rvec) - R).max() < 1e-9
    mask = np.zeros(len(X), bool); mask[inliers] = True
    assert mask[~bad].mean() > 0.95 and mask[bad].mean() < 0.05
