"""what makes the c5 shape slow under a prioritised second stream inside bench.py: python3 tools/experiments/c5_probe2.py <idle_ctxs> <closed_ctxs>"""
import argparse, os, sys
sys.path.insert(0, os.getcwd())
import bench
from eacham_amd import synth, HipContext
idle, closed = int(sys.argv[1]), int(sys.argv[2])
a = argparse.Namespace(gpus=1, backend="nccl", all_on_device=-1, single_process=False)
D = bench.Dist(a)
for _ in range(closed):
    c = HipContext(0); c.sync(); c.close()
mode = os.environ.get("EACHAM_STREAM2_PRIORITY", "high")
os.environ["EACHAM_STREAM2_PRIORITY"] = os.environ.get("IDLE_MODE", mode)   # the idle contexts' second streams
keep = [HipContext(0) for _ in range(idle)]
os.environ["EACHAM_STREAM2_PRIORITY"] = mode
kit = synth.make_scene(100, 15_000, 10, seed=5)
kd, _ = synth.make_frame_descriptors(kit, 1500, 128, seed=5)
for gather in (False, True):
    out, r = bench.matching_line(D, kd, "i8", 128, 5, 2, "c5 probe", bench.sweep_kernel(128), gather_at_one=gather)
    print(os.environ.get("EACHAM_STREAM2_PRIORITY", "default"), "idle", idle, "closed", closed, "gather", gather, "pairs/s", round(out["value"]), "ms/step", round(out["ms_per_step"], 3),
          "sweep ms/step", round(r["tile_ms"] / 5, 3), "fin ms/step", round(r["fin_ms"] / 5, 3), flush=True)
