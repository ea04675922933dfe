"""Generates tests/golden/match_golden.npz with the CPU oracle (oracle/match_oracle.c).

The reference has no golden vectors for this path (parity unpinned), so these are
produced by the repo's own restatement; the file records what generated it.
Run from the repo root:  python tests/golden/make_match_golden.py
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from eacham_amd import synth  # noqa: E402
import oracle_api as O  # noqa: E402


def main():
    scene = synth.make_scene(n_cams=4, n_landmarks=160, k_obs=3, seed=synth.MASTER_SEED)
    descs, ids = synth.make_frame_descriptors(scene, n_kpts=96, dim=128, seed=synth.MASTER_SEED)
    # ragged + tie cases: frame 3 is shortened and gets duplicated rows
    descs[3] = descs[3][:70].copy()
    descs[3][5] = descs[3][4]
    descs[2][10] = descs[2][9]
    pairs = synth.all_pairs(len(descs))
    out = {"pairs": pairs, "dim": np.int32(128), "ratio": np.float64(0.8)}
    for f, d in enumerate(descs):
        out[f"desc{f}"] = d.astype(np.uint8)
    for min_dir, min_mut, tag in [(30, 30, "ref"), (5, 5, "low")]:
        c, o, q, t, st, _ = O.match_all_pairs(descs, pairs, 0.8, min_dir, min_mut)
        out[f"counts_{tag}"], out[f"offsets_{tag}"], out[f"q_{tag}"], out[f"t_{tag}"], out[f"stats_{tag}"] = c, o, q, t, st
    for (a, b) in [(0, 1), (1, 0), (2, 3), (3, 2)]:
        q, t = O.match_directed(descs[a], descs[b])
        out[f"dir_{a}_{b}_q"], out[f"dir_{a}_{b}_t"] = q, t
    rev = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    out["generator"] = np.array(f"oracle/match_oracle.c @ {rev}; exact 2-NN L2, ratio 0.8, thresholds 30/30 and 5/5")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "match_golden.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
