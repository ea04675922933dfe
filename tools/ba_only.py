"""S200 RefineBA solves only (for rocprofv3 --kernel-trace --stats): python3 tools/ba_only.py [solves] [cams] [landmarks] [seed]"""
import os, sys
sys.path.insert(0, os.getcwd())
from eacham_amd import HipContext, synth, ba
solves = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cams = int(sys.argv[2]) if len(sys.argv) > 2 else 200
lms = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
seed = int(sys.argv[4]) if len(sys.argv) > 4 else synth.MASTER_SEED
A = ba.BaArrays.from_scene(synth.make_scene(cams, lms, 10, seed=seed))
ctx = HipContext(0)
s = ba.PreparedBA(ctx, A)
cfg = ba.OptimizerConfig.refine_ba()
n = 0
for _ in range(solves):
    n += s.run(cfg, trace_cap=0).inner_iterations
print("inner iterations", n)
s.close()
