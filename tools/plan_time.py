"""Host time of build_ba_plan (eacham_amd/csrc/ba_plan.hpp) on the camera graphs of S200 and config 4, on THIS machine's
cores: python3 tools/plan_time.py   (compiles tools/experiments/time_plan.cpp with g++)"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eacham_amd import synth
exe = "/tmp/time_plan"
subprocess.run(["g++", "-O3", "-std=c++17", "-o", exe, os.path.join(ROOT, "tools", "experiments", "time_plan.cpp"), "-lpthread"], check=True)
for nc, nl in [(200, 50000), (500, 100000)]:
    sc = synth.make_scene(nc, nl, 10, seed=12345)
    cam, lm = sc["obs_cam"].astype(np.int64), sc["obs_lm"].astype(np.int64)
    order = np.argsort(lm, kind="stable")
    cam, lm = cam[order], lm[order]
    ptr = np.searchsorted(lm, np.arange(nl + 1))
    edges = set()
    for j in range(nl):
        cs = np.unique(cam[ptr[j]:ptr[j + 1]])
        for a in range(len(cs)):
            for b in range(a + 1, len(cs)):
                edges.add((int(cs[a]), int(cs[b])))
    inp = f"{nc} {len(edges)}\n" + "\n".join(f"{a} {b}" for a, b in sorted(edges)) + "\n"
    print(nc, "cameras", len(edges), "edges")
    print(subprocess.run([exe], input=inp, capture_output=True, text=True).stdout)
