"""Triangulation throughput on the S200 scene (50,000 tracks, 2..10 observers each): device kernel
time from the C-ABI's HIP-event slots, end-to-end rate through the host-pointer entry point, and the
CPU oracle beside it. Run on the GPU box:  python tests/rate_tri.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # (this file lives in tests/: it times the CPU oracle beside the device path, which only tests may do)

from eacham_amd import HipContext, capi, synth  # noqa: E402
from eacham_amd import triangulate as tri  # noqa: E402


def main():
    sc = synth.make_scene(200, 50_000, 10, seed=synth.MASTER_SEED)
    tr = synth.make_tracks(sc, seed=1, min_obs=2, outlier_frac=0.15)
    m = np.diff(tr["track_ptr"])
    pairs = int(np.where(m < 2, 0, np.where(m == 2, 1, m * (m - 1) // 2)).sum())
    args = (tr["transforms"], tr["track_ptr"], tr["obs_frame"], tr["obs_uv"], tr["K"], 4.0, 3.0 * 3.141592 / 180.0)
    out = {"tracks": int(m.size), "observations": int(m.sum()), "pairs": pairs}
    with HipContext(0) as ctx:
        tri.triangulate_tracks(ctx, *args)
        ctx.profile_enable(True)
        ctx.profile_reset()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            pts, status, masks = tri.triangulate_tracks(ctx, *args)
        dt = (time.perf_counter() - t0) / reps
        n, ms = ctx.profile_get(capi.KERNEL_TRIANGULATE)
        out.update(kernel_ms=ms / n, end_to_end_ms=dt * 1e3, tracks_per_s_kernel=m.size / (ms / n) * 1e3,
                   tracks_per_s_end_to_end=m.size / dt, pairs_per_s_kernel=pairs / (ms / n) * 1e3,
                   accepted=int((status == 3).sum()))
    import oracle_api as O
    t0 = time.perf_counter()
    opts, ostatus, omasks = O.tri_tracks(*args)
    dt = time.perf_counter() - t0
    out.update(cpu_oracle_ms=dt * 1e3, cpu_tracks_per_s=m.size / dt, cpu_cores=os.cpu_count(),
               parity=bool(np.array_equal(status, ostatus) and np.array_equal(masks, omasks)))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
