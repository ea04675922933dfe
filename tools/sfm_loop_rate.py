"""The incremental loop of apps/sfm/main.cpp on the library (tests/cpp/sfm_loop_driver.cpp) on a longer synthetic sequence:
python3 tools/sfm_loop_rate.py [frames] [kpts] [landmarks] [k_obs]   (TUM stand-in sizes: 500 600 30000 10)"""
import json, os, struct, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eacham_amd import synth  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 100
kpts = int(sys.argv[2]) if len(sys.argv) > 2 else 600
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 6000
kobs = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dim = 128
sc = synth.make_scene(F, nl, kobs, seed=3, pixel_noise=0.5)
descs, ids = synth.make_frame_descriptors(sc, kpts, dim, seed=3)
uv_of = {}
for o, (c, l) in enumerate(zip(sc["obs_cam"].tolist(), sc["obs_lm"].tolist())):
    uv_of[(c, l)] = sc["obs_uv"][o]
rnd = synth.rng_uniform(3, 900, (F, kpts, 2)) * 800.0
kp = np.array([[uv_of.get((f, int(ids[f][k])), rnd[f, k]) for k in range(kpts)] for f in range(F)], dtype=np.float32)
K = sc["K"]
tmp = tempfile.mkdtemp()
exe = os.path.join(tmp, "sfm_loop_driver")
lib = os.path.join(ROOT, "eacham_amd", "lib")
cpp = os.path.join(ROOT, "tests", "cpp")
subprocess.run(["g++", "-std=c++17", "-O2", *(["-DEACHAM_RECON_DEBUG"] if os.environ.get("SFM_DEBUG") else []), "-I" + os.path.join(ROOT, "include"), "-I" + cpp, os.path.join(cpp, "sfm_loop_driver.cpp"), "-o", exe,
                "-L" + lib, "-leacham_hip", "-Wl,-rpath," + lib, "-lpthread"], check=True)
fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
deg = 3.141592 / 180.0
with open(fin, "wb") as f:
    f.write(struct.pack("ii", F, dim))
    for fr in range(F):
        f.write(struct.pack("i", kpts)); f.write(kp[fr].tobytes()); f.write(descs[fr].tobytes())
    f.write(np.array([K[0], 0, K[2], 0, K[1], K[3], 0, 0, 1.0]).tobytes())
    f.write(np.array([0.8, 3.5, 3.0 * deg, 8.0, 3.0 * deg, 15, 100, float(os.environ.get('SFM_INLIER_PX', '4.0'))], dtype=np.float32).tobytes())
t0 = time.perf_counter()
r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=1000)
dt = time.perf_counter() - t0
print(r.stdout.strip()[-600:], r.stderr.strip()[:1500])
if r.returncode == 0:
    def vec(f):
        n = struct.unpack("q", f.read(8))[0]
        return np.frombuffer(f.read(8 * n), dtype=np.float64).copy()
    with open(fout, "rb") as f:
        log, poses, pts = vec(f), vec(f).reshape(F, 17), vec(f).reshape(-1, 6)
    valid = poses[:, 0] == 1
    T = poses[:, 1:].reshape(F, 4, 4)
    Ce = np.array([-T[i, :3, :3].T @ T[i, :3, 3] for i in range(F)])
    Tt = sc["T_true"]
    Ct = np.array([-Tt[i, :3, :3].T @ Tt[i, :3, 3] for i in range(F)])
    a, b = Ce[valid] - Ce[valid].mean(0), Ct[valid] - Ct[valid].mean(0)
    U, S, Vt = np.linalg.svd(b.T @ a / valid.sum())
    D = np.eye(3); D[2, 2] = np.sign(np.linalg.det(U) * np.linalg.det(Vt))
    R = U @ D @ Vt
    s = np.trace(np.diag(S) @ D) / (a ** 2).sum() * valid.sum()
    err = np.linalg.norm((s * (R @ a.T)).T - b, axis=1)
    if os.environ.get("SFM_DEBUG"):
        sel = np.flatnonzero(valid)[:10]                        # alignment on the first ten frames only: where does it leave the truth?
        a2, b2 = Ce[sel] - Ce[sel].mean(0), Ct[sel] - Ct[sel].mean(0)
        U2, S2, Vt2 = np.linalg.svd(b2.T @ a2 / len(sel))
        D2 = np.eye(3); D2[2, 2] = np.sign(np.linalg.det(U2) * np.linalg.det(Vt2))
        R2 = U2 @ D2 @ Vt2
        s2 = np.trace(np.diag(S2) @ D2) / (a2 ** 2).sum() * len(sel)
        e2 = np.linalg.norm((s2 * (R2 @ (Ce - Ce[sel].mean(0)).T)).T - (Ct - Ct[sel].mean(0)), axis=1)
        print("err by frame (aligned on first 10):", np.round(e2, 2).tolist())
    print(json.dumps({"frames": F, "valid": int(valid.sum()), "centre_err_max": float(err.max()), "centre_err_median": float(np.median(err)),
                      "map_points": int(len(pts)), "valid_points": int((pts[:, 4] == 1).sum()), "process_wall_s": round(dt, 2)}))
