"""Wall time per LM inner iteration of S200 RefineBA solves: python3 tools/ba_wall.py [solves] [cams] [landmarks]
(EACHAM_HIP_LIB selects the library build; used for same-box A/B runs)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
from eacham_amd import HipContext, synth, ba
solves = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cams = int(sys.argv[2]) if len(sys.argv) > 2 else 200
lms = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
A = ba.BaArrays.from_scene(synth.make_scene(cams, lms, 10))
ctx = HipContext(0)
s = ba.PreparedBA(ctx, A)
cfg = ba.OptimizerConfig.refine_ba()
for _ in range(3):
    s.run(cfg, trace_cap=0)
best = 1e9
tot_t = tot_n = 0
for _ in range(solves):
    t0 = time.perf_counter()
    r = s.run(cfg, trace_cap=0)
    dt = time.perf_counter() - t0
    tot_t += dt; tot_n += r.inner_iterations
    best = min(best, dt / r.inner_iterations)
print(f"{os.environ.get('EACHAM_HIP_LIB', 'default')}: {1e3 * tot_t / tot_n:.4f} ms per inner iteration (best solve {1e3 * best:.4f}), {tot_n} iterations")
s.close()
