"""Rate of the drop-in FeatureMatcherHip::Match under the reference's own call pattern (apps/sfm/main.cpp:84-109: one
std::async(&Match) per ORDERED pair from a pool of threads on one shared instance), next to the batch entry point.
  python3 tools/match_async_rate.py [frames] [kpts] [dim] [threads]"""
import os, struct, subprocess, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
from eacham_amd import synth, HipContext

F = int(sys.argv[1]) if len(sys.argv) > 1 else 48
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
D = int(sys.argv[3]) if len(sys.argv) > 3 else 128
T = int(sys.argv[4]) if len(sys.argv) > 4 else 16
root = os.getcwd()
sc = synth.make_scene(F, F * N // 10, 10, seed=9)
descs, _ = synth.make_frame_descriptors(sc, N, D, seed=9)
tmp = tempfile.mkdtemp()
fin, fout, exe = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin"), os.path.join(tmp, "drv")
with open(fin, "wb") as f:
    f.write(struct.pack("ii", F, D))
    for d in descs:
        f.write(struct.pack("i", d.shape[0])); f.write(np.ascontiguousarray(d, np.float32).tobytes())
lib = os.path.join(root, "eacham_amd", "lib")
subprocess.run(["g++", "-std=c++17", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "match_async_driver.cpp"),
                "-o", exe, "-L" + lib, "-leacham_hip", "-Wl,-rpath," + lib, "-lpthread"], check=True)
for threads in sorted({1, 4, T}):
    r = subprocess.run([exe, fin, fout, str(threads), "3"], capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr.strip())
# the batch entry point on the same frames (unordered pairs: one unit = both directions + mutual check)
ctx = HipContext(0)
for f, d in enumerate(descs):
    ctx.upload_descriptors(f, d)
pairs = synth.all_pairs(F)
ctx.match_all_pairs(pairs)
t0 = time.perf_counter(); ctx.match_all_pairs(pairs); dt = time.perf_counter() - t0
print(f"eacham_match_all_pairs (host-pointer CSR form): {len(pairs)} unordered pairs in {dt*1e3:.1f} ms = {len(pairs)/dt:.0f} pairs/s "
      f"= {2*len(pairs)/dt:.0f} directed Match()-equivalents/s")
