"""Wall time per LM inner iteration under every elimination ordering: python3 tools/ba_orderings.py [solves] [cams] [landmarks] [seed]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
from eacham_amd import HipContext, synth, ba, capi
solves = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cams = int(sys.argv[2]) if len(sys.argv) > 2 else 200
lms = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
seed = int(sys.argv[4]) if len(sys.argv) > 4 else synth.MASTER_SEED
A = ba.BaArrays.from_scene(synth.make_scene(cams, lms, 10, seed=seed))
ctx = HipContext(0)
cfg = ba.OptimizerConfig.refine_ba()
for name in ("natural", "rcm", "nd", "auto"):
    A.ordering = name
    t0 = time.perf_counter()
    s = ba.PreparedBA(ctx, A)
    prep = time.perf_counter() - t0
    for _ in range(2):
        s.run(cfg, trace_cap=0)
    tot_t = tot_n = 0
    for _ in range(solves):
        t0 = time.perf_counter()
        r = s.run(cfg, trace_cap=0)
        tot_t += time.perf_counter() - t0
        tot_n += r.inner_iterations
    ctx.profile_reset(); ctx.profile_enable(True)
    n = 0
    for _ in range(3):
        n += s.run(cfg, trace_cap=0).inner_iterations
    ctx.profile_enable(False)
    st = {k: ctx.profile_get(i)[1] / n * 1e3 for k, i in (("lin", capi.KERNEL_BA_LINEARIZE), ("schur", capi.KERNEL_BA_SCHUR), ("solve", capi.KERNEL_BA_SOLVE), ("err", capi.KERNEL_BA_ERROR))}
    print(f"{name:8s} {1e3 * tot_t / tot_n:.4f} ms/inner  final {r.final_error:.6e} iters {r.outer_iterations}/{r.inner_iterations}  prepare {1e3 * prep:.1f} ms  "
          + " ".join(f"{k} {v:.1f}us" for k, v in st.items()) + f"  plan {s.plan_info()}", flush=True)
    s.close()
