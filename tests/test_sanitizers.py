"""CPU: host-side sanitizer runs ("TSAN/ASAN on the host restatement in CI here").

  * the C oracle (oracle/*.c) under AddressSanitizer + UndefinedBehaviorSanitizer: its own test files are re-run in a
    child python that preloads libasan and loads the sanitized build;
  * the header-only C++ adapters (include/eacham/*.hpp) under ASAN + UBSAN, and the concurrent drop-in Match() under
    ThreadSanitizer, linked against tests/cpp/stub_abi.cpp — a CPU stand-in for the C-ABI that computes a fake,
    content-dependent match rule, so the run also checks that every caller gets the result of ITS pair and that no
    cached upload is served for other values.
GPU AddressSanitizer is not available on this pool: device code is covered by the parity tests only."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
ASAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
TSAN = ["-fsanitize=thread"]
SAN_ENV = dict(ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=66", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               TSAN_OPTIONS="halt_on_error=1:exitcode=66")


def _gcc_file(name):
    return subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()


def _build(tmp, driver, flags, tag):
    exe = os.path.join(tmp, f"{driver}_{tag}")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-Werror", *flags, "-I" + os.path.join(ROOT, "include"), "-I" + CPP,
           os.path.join(CPP, driver + ".cpp"),
           os.path.join(CPP, "stub_abi.cpp"), "-o", exe, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def _run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **SAN_ENV), timeout=600, **kw)
    assert r.returncode == 0 and "ERROR: " not in r.stderr and "runtime error" not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr, \
        (r.returncode, r.stdout[-1500:], r.stderr[-4000:])
    return r


def test_oracle_under_asan_ubsan():
    import oracle
    so = oracle.build_sanitized()
    libasan = _gcc_file("libasan.so")
    assert os.path.isabs(libasan), "gcc has no libasan.so"
    env = dict(os.environ, LD_PRELOAD=libasan, EACHAM_ORACLE_LIB=so, OMP_NUM_THREADS="4",
               ASAN_OPTIONS="detect_leaks=0:exitcode=66", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    files = [os.path.join(ROOT, "tests", f) for f in ("test_match_oracle.py", "test_ba_oracle.py", "test_tri_oracle.py", "test_graph_oracle.py")]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", *files], env=env, capture_output=True,
                       text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, \
        (r.returncode, r.stdout[-2500:], r.stderr[-4000:])
    assert " passed" in r.stdout


def _fake_rule(A, B):
    """tests/cpp/stub_abi.cpp fake_match."""
    if A.shape[0] <= 0 or B.shape[0] < 2:
        return {}
    b0 = int(np.rint(abs(16.0 * float(B[0, 0]))))
    out = {}
    for i in range(A.shape[0]):
        a0, a1 = int(np.rint(abs(16.0 * float(A[i, 0])))), int(np.rint(abs(16.0 * float(A[i, -1]))))
        if (a0 + b0) % 3:
            out[i] = (a1 + i) % B.shape[0]
    return out


@pytest.mark.parametrize("tag,flags", [("asan", ASAN), ("tsan", TSAN)])
@pytest.mark.parametrize("kind", ["u8", "f32"])
def test_concurrent_drop_in_match_under_sanitizers(tmp_path, tag, flags, kind):
    """FeatureMatcherHip::Match from 8 threads, one std::async per ordered pair (apps/sfm/main.cpp:98-109)."""
    from eacham_amd import synth
    tmp = str(tmp_path)
    exe = _build(tmp, "match_async_driver", flags, tag)
    F, dim = 9, 32
    if kind == "u8":
        descs = [synth.random_u8_descriptors(40 + 7 * f, dim, 77, f) for f in range(F)]
        descs[4] = descs[4][:1]
        descs[6] = descs[6][:0]
    else:
        descs = [synth.unit_float_descriptors(40 + 7 * f, dim, 77, f) for f in range(F)]
    fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("ii", F, dim))
        for d in descs:
            f.write(struct.pack("i", d.shape[0])); f.write(np.ascontiguousarray(d, np.float32).tobytes())
    _run([exe, fin, fout, "8", "3"])
    with open(fout, "rb") as f:
        for i in range(F):
            for j in range(F):
                if i == j:
                    continue
                n = struct.unpack("q", f.read(8))[0]
                got = np.frombuffer(f.read(4 * n), dtype=np.uint32).reshape(-1, 2)
                assert dict(map(tuple, got.tolist())) == _fake_rule(descs[i], descs[j]), (i, j)
        seconds, calls, batches, uploads, hits = np.frombuffer(f.read(40), dtype=np.float64)
    assert calls == 3 * F * (F - 1) and uploads == (F if kind == "u8" else F + 1) and batches <= calls


def test_adapters_under_asan_ubsan(tmp_path):
    """The graph / map walks and write-backs of BundleAdjusterHip.hpp, TriangulatorHip.hpp, FeatureMatcherHip.hpp and the
    JSON reader / writers of SfmIO.hpp on the inputs of the GPU adapter tests (the stub passes values through)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_cpp_adapters as T
    tmp = str(tmp_path)
    fx = T.write_adapter_fixture(tmp)
    r = _run([_build(tmp, "adapter_driver", ASAN, "asan"), fx.fin, fx.fout])
    assert "adapter driver ok" in r.stdout
    tx = T.write_tri_fixture(tmp)
    _run([_build(tmp, "tri_driver", ASAN, "asan"), tx.fin, tx.fout])
    # The reference-typed entry points (include/eacham/ReferenceGlue.hpp: RefineBA(int, shared_ptr<graph_t>, shared_ptr<Map>,
    # cv::Mat&, const OptimizerConfig&) and TriangulateFrame(...), BundleAdjuster.h:13-17 / Triangulator.h:41-43) compiled
    # against stand-ins that carry the accessor names of modules/sfm/data/{Graph,Node,Map}.h (tests/cpp/ref_standins.hpp):
    # filling the views from the objects, the call and the write-back into the objects must give, byte for byte, what the
    # drivers get with hand-filled views.
    for driver, fixture in (("adapter_driver", fx), ("tri_driver", tx)):
        plain = open(fixture.fout, "rb").read()
        _run([_build(tmp, driver, ASAN + ["-DEACHAM_TEST_GLUE"], "glue_asan"), fixture.fin, fixture.fout])
        assert open(fixture.fout, "rb").read() == plain and len(plain) > 1000, driver
    # SfmIO.hpp needs no ABI at all
    exe = os.path.join(tmp, "io_driver_asan")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", *ASAN, "-I" + os.path.join(ROOT, "include"), os.path.join(CPP, "io_driver.cpp"), "-o", exe],
                   check=True, capture_output=True)
    import shutil
    for name in ("ref_SfmConfig.json", "ref_SfmConfigNerf.json"):
        d = os.path.join(tmp, name + ".d")
        os.mkdir(d)
        shutil.copy(os.path.join(ROOT, "tests", "golden", name), os.path.join(d, "config.json"))
        _run([exe, d, "config-only"])
    bad = os.path.join(tmp, "bad.d")
    os.mkdir(bad)
    open(os.path.join(bad, "config.json"), "w").write('{"root_path": "x", "feature": {"min_features_count": [1, 2, {"a": "\\u00e9\\n"}]}, "t": tru')
    r = subprocess.run([exe, bad, "config-only"], capture_output=True, text=True, env=dict(os.environ, **SAN_ENV))
    assert r.returncode == 1 and "ERROR: " not in r.stderr and "runtime error" not in r.stderr   # a parse error, reported, no UB


def test_two_view_and_pnp_adapters_under_asan_ubsan(tmp_path):
    """TwoViewHip.hpp / PnPHip.hpp — the LMedS and RANSAC loops, the sample generator, the homography refit, the
    decompositions and the Rodrigues vector — with the stub standing in for the device calls (it reads every input array it
    is handed and fills every output array: a wrong size in the adapters is an ASAN report)."""
    import struct
    tmp = str(tmp_path)
    exe = _build(tmp, "twoview_driver", ASAN, "asan")
    rng = np.random.default_rng(4)
    K = np.array([[700.0, 0, 320], [0, 700, 240], [0, 0, 1]])
    H = K @ (np.eye(3) + 0.1 * rng.normal(size=(3, 3))) @ np.linalg.inv(K)
    lines = ["H " + " ".join(f"{x:.17g}" for x in np.r_[H.ravel(), K.ravel()]),
             "H " + " ".join(f"{x:.17g}" for x in np.r_[np.eye(3).ravel(), K.ravel()]),       # no motion at all
             "H " + " ".join(f"{x:.17g}" for x in np.r_[np.zeros(9), K.ravel()]),             # not a homography
             "E " + " ".join(f"{x:.17g}" for x in rng.normal(size=9)), "E " + " ".join(["0"] * 9),
             "R " + " ".join(f"{x:.17g}" for x in np.eye(3).ravel()), "R " + " ".join(f"{x:.17g}" for x in np.diag([1.0, -1, -1]).ravel())]
    r = _run([exe, "decompose"], input="\n".join(lines) + "\n")
    assert r.stdout.count("\n") >= 7
    fin, fout = os.path.join(tmp, "tv_in.bin"), os.path.join(tmp, "tv_out.bin")
    with open(fin, "wb") as f:
        for n in (37, 5):                                           # two scenes; the second at the five-point minimum
            f.write(struct.pack("i", n))
            f.write(rng.uniform(0, 640, size=(n, 2)).tobytes()); f.write(rng.uniform(0, 640, size=(n, 2)).tobytes())
            f.write(K.tobytes())
    assert "twoview driver ok" in _run([exe, "pipeline", fin, fout]).stdout
    for n in (300, 5, 3):                                            # more than a chunk's worth of points, the minimum, too few
        with open(fin, "wb") as f:
            f.write(struct.pack("i", n))
            f.write(rng.normal(size=(n, 3)).tobytes()); f.write(rng.uniform(0, 640, size=(n, 2)).tobytes()); f.write(K.tobytes())
        assert "twoview driver ok" in _run([exe, "pnp", fin, fout]).stdout


def test_sfm_loop_driver_builds_and_walks_its_graph_under_asan_ubsan(tmp_path):
    """tests/cpp/sfm_loop_driver.cpp (the loop of apps/sfm/main.cpp on ReferenceGlue.hpp + ReconstructionHip.hpp) compiled
    -Wall -Wextra -Werror against the stub: the graph / map walks of FindBestPair, RecoverPoseTwoView and RecoverPosePnP run
    on fake device results; whichever way the loop ends (a reconstruction or 'no initial pair'), there is no sanitizer report."""
    tmp = str(tmp_path)
    exe = _build(tmp, "sfm_loop_driver", ASAN, "asan")
    rng = np.random.default_rng(9)
    F, n, dim = 4, 60, 16
    fin, fout = os.path.join(tmp, "sfm_in.bin"), os.path.join(tmp, "sfm_out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("ii", F, dim))
        base = rng.integers(0, 255, size=(n, dim)).astype(np.float32)
        for _ in range(F):
            f.write(struct.pack("i", n)); f.write(rng.uniform(0, 640, size=(n, 2)).astype(np.float32).tobytes()); f.write(base.tobytes())
        f.write(np.array([700.0, 0, 320, 0, 700, 240, 0, 0, 1]).tobytes())
        f.write(np.array([0.8, 3.5, 0.05, 8.0, 0.05, 15, 10], dtype=np.float32).tobytes())
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, env=dict(os.environ, **SAN_ENV), timeout=600)
    assert r.returncode in (0, 3) and "ERROR: " not in r.stderr and "runtime error" not in r.stderr, (r.returncode, r.stdout[-1000:], r.stderr[-3000:])
    assert "sfm loop" in r.stdout
