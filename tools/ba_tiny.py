"""S-sized-down RefineBA solves for kernel-floor measurements under rocprofv3: python3 tools/ba_tiny.py <cams> <landmarks>
(with 8 cameras / 64 landmarks every per-landmark kernel is ONE wave: what it still costs is launch + the length of a thread's
instruction stream, the reading behind BaDev::lpl)."""
import os, sys
sys.path.insert(0, os.getcwd())
from eacham_amd import HipContext, synth, ba
cams = int(sys.argv[1]); lms = int(sys.argv[2])
A = ba.BaArrays.from_scene(synth.make_scene(cams, lms, min(cams, 8)))
ctx = HipContext(0); s = ba.PreparedBA(ctx, A); cfg = ba.OptimizerConfig.refine_ba()
n = 0
for _ in range(20): n += s.run(cfg, trace_cap=0).inner_iterations
print("inner iterations", n)
