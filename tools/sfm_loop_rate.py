"""The incremental loop of apps/sfm/main.cpp on the library (tests/cpp/sfm_loop_driver.cpp) on a longer synthetic sequence:
python3 tools/sfm_loop_rate.py [frames] [kpts] [landmarks] [k_obs]   (TUM stand-in sizes: 500 600 30000 10)
SFM_INLIER_PX (default 4.0; 0 = the reference's LMedS masks for the H / E branch), SFM_DEBUG=1 for the two-view log.
bench.py imports run() for the `sfm_loop` part of its c3_tum line."""
import json, os, re, struct, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(F=100, kpts=600, nl=6000, kobs=10, inlier_px=4.0, debug=False, seed=3):
    from eacham_amd import synth
    dim = 128
    sc = synth.make_scene(F, nl, kobs, seed=seed, pixel_noise=0.5)
    descs, ids = synth.make_frame_descriptors(sc, kpts, dim, seed=seed)
    uv_of = {}
    for o, (c, l) in enumerate(zip(sc["obs_cam"].tolist(), sc["obs_lm"].tolist())):
        uv_of[(c, l)] = sc["obs_uv"][o]
    rnd = synth.rng_uniform(seed, 900, (F, kpts, 2)) * 800.0
    kp = np.array([[uv_of.get((f, int(ids[f][k])), rnd[f, k]) for k in range(kpts)] for f in range(F)], dtype=np.float32)
    K = sc["K"]
    tmp = tempfile.mkdtemp()
    exe = os.path.join(tmp, "sfm_loop_driver")
    lib = os.path.join(ROOT, "eacham_amd", "lib")
    cpp = os.path.join(ROOT, "tests", "cpp")
    t_cc = time.perf_counter()
    subprocess.run(["g++", "-std=c++17", "-O2", *(["-DEACHAM_RECON_DEBUG", "-DEACHAM_GLUE_TIMING"] if debug else []), "-I" + os.path.join(ROOT, "include"), "-I" + cpp,
                    os.path.join(cpp, "sfm_loop_driver.cpp"), "-o", exe, "-L" + lib, "-leacham_hip", "-Wl,-rpath," + lib, "-lpthread"], check=True)
    compile_s = time.perf_counter() - t_cc   # the host compile of the driver: reported apart from the loop
    fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
    deg = 3.141592 / 180.0
    with open(fin, "wb") as f:
        f.write(struct.pack("ii", F, dim))
        for fr in range(F):
            f.write(struct.pack("i", kpts)); f.write(kp[fr].tobytes()); f.write(descs[fr].tobytes())
        f.write(np.array([K[0], 0, K[2], 0, K[1], K[3], 0, 0, 1.0]).tobytes())
        f.write(np.array([0.8, 3.5, 3.0 * deg, 8.0, 3.0 * deg, 15, 100, inlier_px], dtype=np.float32).tobytes())
    if os.environ.get("SFM_PREPARE_ONLY"):   # leave the driver and its input for a profiler run: `rocprofv3 ... -- <exe> <in> <out>`
        print(exe, fin, fout)
        sys.exit(0)
    t0 = time.perf_counter()
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=1000)
    dt = time.perf_counter() - t0
    out = {"frames": F, "kpts": kpts, "landmarks": nl, "returncode": r.returncode, "driver": r.stdout.strip()[-700:], "process_wall_s": round(dt, 2), "driver_compile_s": round(compile_s, 2)}
    if debug:
        out["stderr"] = r.stderr.strip()[:1500]
    if r.returncode != 0:
        return out

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
    m = re.search(r"\[Match\] ([0-9.]+) ms .*\[SfM\] ([0-9.]+) ms = PnP ([0-9.]+) \+ TriangulateFrame ([0-9.]+) \+ RefineBA ([0-9.]+) \+ GetBestPairForValid ([0-9.]+) \+ global BA ([0-9.]+)", r.stdout)
    if m:
        match_ms, sfm_ms, pnp, tri, rba, query, gba = (float(x) for x in m.groups())
        added = max(int(valid.sum()) - 2, 1)
        out.update(match_ms=match_ms, sfm_ms=sfm_ms, frames_per_s=round(added / (sfm_ms * 1e-3), 1),
                   ms_per_frame={"pnp": round(pnp / added, 2), "triangulate_frame_x2": round(tri / added, 2), "refine_ba": round(rba / added, 2),
                                 "next_pair_query": round(query / added, 2)}, global_ba_ms=gba)
    out.update(valid=int(valid.sum()), centre_err_max=float(err.max()), centre_err_median=float(np.median(err)), map_points=int(len(pts)),
               valid_points=int((pts[:, 4] == 1).sum()))
    return out


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:5]]
    defaults = [100, 600, 6000, 10]
    F, kpts, nl, kobs = a + defaults[len(a):]
    res = run(F, kpts, nl, kobs, float(os.environ.get("SFM_INLIER_PX", "4.0")), bool(os.environ.get("SFM_DEBUG")))
    print(res.pop("driver"))
    if "stderr" in res:
        print(res.pop("stderr"))
    print(json.dumps(res))
