"""c5_kitti shape with / without the one-rank all-gather, for the stream-priority A/B: python3 tools/experiments/c5_probe.py"""
import argparse, os, sys
sys.path.insert(0, os.getcwd())
import bench
from eacham_amd import synth
a = argparse.Namespace(gpus=1, backend="nccl", all_on_device=-1, single_process=False)
D = bench.Dist(a)
kit = synth.make_scene(100, 15_000, 10, seed=5)
kd, _ = synth.make_frame_descriptors(kit, 1500, 128, seed=5)
for gather in (False, True, False, True):
    out, r = bench.matching_line(D, kd, "i8", 128, 5, 2, "c5 probe", bench.sweep_kernel(128), gather_at_one=gather)
    print(os.environ.get("EACHAM_STREAM2_PRIORITY", "default"), "gather", gather, "pairs/s", round(out["value"]), "ms/step", round(out["ms_per_step"], 3),
          "sweep ms/step", round(r["tile_ms"] / 5, 3), "fin ms/step", round(r["fin_ms"] / 5, 3), flush=True)
