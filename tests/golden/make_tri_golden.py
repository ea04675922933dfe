"""Generates tests/golden/tri_golden.npz with the CPU oracle (oracle/tri_oracle.c).

The reference has no tests or vectors for triangulation and Eigen is absent here (parity
unpinned), so the fixture records the repo's own restatement, cross-checked in
tests/test_tri_oracle.py against a literal numpy/LAPACK statement of Triangulator.cpp:96-186.
Run from the repo root:  python tests/golden/make_tri_golden.py
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

MAX_ERR = 4.0                   # config/SfmConfig.json:18
MIN_ANGLE = 3.0 * 3.141592 / 180.0  # config/SfmConfig.json:19 through SfmConfig.h:52-53


def main():
    sc = synth.make_scene(24, 400, 6, seed=synth.MASTER_SEED, pixel_noise=1.0)
    tr = synth.make_tracks(sc, seed=7, min_obs=1, outlier_frac=0.2)
    pts, status, masks = O.tri_tracks(tr["transforms"], tr["track_ptr"], tr["obs_frame"], tr["obs_uv"], tr["K"], MAX_ERR, MIN_ANGLE)
    rev = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tri_golden.npz"), transforms=tr["transforms"],
                        track_ptr=tr["track_ptr"], obs_frame=tr["obs_frame"], obs_uv=tr["obs_uv"], K=tr["K"],
                        max_err=np.float32(MAX_ERR), min_angle=np.float32(MIN_ANGLE), points=pts, status=status, masks=masks,
                        generator=np.array(f"oracle/tri_oracle.c @ {rev}; SfmConfig.json thresholds"))
    print("tracks", status.size, "status histogram", np.bincount(status, minlength=4))


if __name__ == "__main__":
    main()
