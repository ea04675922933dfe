import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from eacham_amd import HipContext, score
import score_cases as SC
rng = np.random.default_rng(1)
n = 2000
tv = SC.two_view_case(n=n, seed=5, outliers=0.25)
sE = np.array([rng.choice(n, 5, replace=False) for _ in range(1000)], np.int32)
def t(fn, reps=10):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    return (time.perf_counter() - t0) / reps * 1e3, r
with HipContext(0) as ctx:
    for ns in (89, 300, 1000):
        ms, (m, c) = t(lambda: score.solve_minimal(ctx, "essential5", tv["uv1"], tv["uv2"], sE[:ns], tv["K"]))
        cand = m[np.arange(m.shape[1])[None, :] < np.asarray(c)[:, None]]
        ms2, _ = t(lambda: score.score_hypotheses(ctx, "essential", tv["uv1"], tv["uv2"], cand, tv["K"], threshold=1e-5, want_errors=False))
        print(ns, "solve_minimal ms", round(ms, 3), "candidates", len(cand), "score ms", round(ms2, 3))
