import sys, time, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from eacham_amd import HipContext, synth
import oracle_api as O
import os
dim=128; n=2000; F=24
base = synth.random_u8_descriptors(n, dim, 7, 0)
descs=[np.clip(base + np.rint(3*synth.rng_normal(7, 10+k, (n,dim))),0,255).astype(np.float32) for k in range(F)]   # every row matches in every pair
pairs = synth.all_pairs(F)
for form in ("exact","bound"):
    os.environ["EACHAM_MATCH_SWEEP_FORM"]=form
    ctx=HipContext(0)
    for f,d in enumerate(descs): ctx.upload_descriptors(f,d)
    got=ctx.match_all_pairs(pairs, stats=False)
    ctx.sync(); t0=time.perf_counter()
    for _ in range(5): got=ctx.match_all_pairs(pairs, stats=False)
    ctx.sync(); dt=(time.perf_counter()-t0)/5
    print(form, "ms per job", round(dt*1e3,2), "matches per pair", got[0].mean())
    if form=="bound":
        want=O.match_all_pairs(descs[:6], synth.all_pairs(6))
        g6=ctx.match_all_pairs(synth.all_pairs(6), stats=False)
        print("bit exact on 15 pairs:", all(np.array_equal(a,b) for a,b in zip(g6[:4], want[:4])))
    ctx.close()
