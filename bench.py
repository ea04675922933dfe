#!/usr/bin/env python3
"""bench.py — headline benchmark of the eacham hot path on MI355X (contract in the task brief).

Metric (BASELINE.json): image-pairs matched/s (+ BA iters/s) on the 200-frame / 50k-landmark
synthetic scene S200. One "step" = one pass of the matching hot path over ALL 19,900 unordered
frame pairs of S200 (2000 keypoints x 256-D per frame): int8-MFMA distance tiles with fused
row/column top-2, ratio test, mutual cross-check, CSR compaction — and, for N > 1, the RCCL
all-gather of the match graph. Inputs are resident in HBM before the timed region.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

The CPU oracle (oracle/) is used here only for the `cpu_baseline` leg (rank 0, N=1, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

I8_MFMA_PEAK_TOPS = 5000.0  # dense int8 MFMA: 2x the ~2.5 PF bf16 dense rate (MI355X_MICROARCH.md, Matrix cores)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--kpts", type=int, default=2000)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--landmarks", type=int, default=50_000)
    ap.add_argument("--cpu-pairs", type=int, default=-1, help="pairs in the CPU-baseline sample (-1 = 4 per core, 0 = skip)")
    ap.add_argument("--ba-solves", type=int, default=5, help="timed RefineBA solves of the S200 window (0 = skip BA)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N > 1 on one GPU)")
    ap.add_argument("--all-on-device", type=int, default=-1, help="rehearsal: put every rank on this device index")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if args.all_on_device >= 0:
        local = args.all_on_device
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from eacham_amd import HipContext, synth, capi, shard

    # ---- synthetic S200 inputs, identical on every rank ------------------------------------
    t_gen = time.time()
    scene = synth.make_scene(args.frames, args.landmarks, 10)
    descs, _ = synth.make_frame_descriptors(scene, args.kpts, args.dim)
    pairs_all = synth.all_pairs(args.frames)
    npairs_total = len(pairs_all)
    # pairs ordered by train frame (L2 reuse of the B operand), contiguous shard per rank
    pairs_all = shard.order_pairs(pairs_all)
    pairs = shard.shard_pairs(pairs_all, world, rank)
    npairs = len(pairs)
    shard_max = shard.shard_capacity(npairs_total, world)
    t_gen = time.time() - t_gen

    ctx = HipContext(local)
    for f, d in enumerate(descs):  # replicated descriptor store: 200 x 2000 x 256 B = 102 MB int8
        ctx.upload_descriptors(f, d)
    ext = torch.cuda.ExternalStream(ctx.stream, device=dev)

    with torch.cuda.stream(ext):
        pairs_dev = torch.from_numpy(pairs).to(dev)
        counts = torch.zeros(shard_max, dtype=torch.int32, device=dev)
        offsets = torch.zeros(npairs + 1, dtype=torch.int64, device=dev)
        total = torch.zeros(1, dtype=torch.int64, device=dev)
        # size the edge buffer from one untimed pass (deterministic inputs -> exact)
        probe_cap = npairs * args.kpts
        edges = torch.zeros(max(probe_cap, 1) * 2, dtype=torch.int32, device=dev)
        ctx.match_all_pairs_dev(pairs_dev.data_ptr(), npairs, counts.data_ptr(), offsets.data_ptr(),
                                edges.data_ptr(), probe_cap, total.data_ptr())
        ctx.sync()
        my_total = int(total.item())
        cap_t = torch.tensor([my_total], dtype=torch.int64, device=dev)
        if world > 1:
            dist.all_reduce(cap_t, op=dist.ReduceOp.MAX)
        edge_cap = max(int(cap_t.item()), 1)
        # two output sets: the all-gather of step i (its own RCCL stream) overlaps the matching of step i+1
        sets = []
        for _ in range(2 if world > 1 else 1):
            st = {"counts": torch.zeros(shard_max, dtype=torch.int32, device=dev),
                  "edges": torch.zeros(edge_cap * 2, dtype=torch.int32, device=dev), "pending": []}
            if world > 1:
                st["g_counts"] = torch.zeros(world * shard_max, dtype=torch.int32, device=dev)
                st["g_edges"] = torch.zeros(world * edge_cap * 2, dtype=torch.int32, device=dev)
            sets.append(st)
    step_no = [0]

    def step():
        st = sets[step_no[0] % len(sets)]
        step_no[0] += 1
        with torch.cuda.stream(ext):
            for w in st["pending"]:  # the exchange that last used this set must be done before it is overwritten
                w.wait()
            st["pending"] = []
            ctx.match_all_pairs_dev(pairs_dev.data_ptr(), npairs, st["counts"].data_ptr(), offsets.data_ptr(),
                                    st["edges"].data_ptr(), edge_cap, total.data_ptr())
            if world > 1:  # RCCL all-gather of the match graph (counts + padded edge lists) over xGMI
                _, _, st["pending"] = shard.all_gather_match_graph(st["counts"], st["edges"], shard_max, edge_cap, world,
                                                                   st["g_counts"], st["g_edges"], async_op=True)

    def fence():
        with torch.cuda.stream(ext):
            for st in sets:
                for w in st["pending"]:
                    w.wait()
                st["pending"] = []
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.profile_reset()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    ctx.profile_enable(False)
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    launches, tile_ms = ctx.profile_get(capi.KERNEL_MATCH_TILE)
    _, fin_ms = ctx.profile_get(capi.KERNEL_MATCH_FINALIZE)
    n_matches = int(total.item())

    ba_out = bench_ba(ctx, scene, args, rank, world, dev) if args.ba_solves > 0 else None

    if rank == 0:
        ops_per_pair = 2.0 * args.kpts * args.kpts * args.dim  # 2*N1*N2*D per unordered pair
        achieved = ops_per_pair * npairs * args.steps / (tile_ms * 1e-3) / 1e12 if tile_ms > 0 else 0.0
        out = {
            "metric": "image-pairs matched/s + BA iters/s, 200-frame/50k-landmark synthetic",
            "value": npairs_total * args.steps / elapsed,
            "unit": "image-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "i8",
            "data": "synthetic",
            "config": {"workload": f"S200 matching: {args.frames} frames x {args.kpts} kpts x {args.dim}-D, "
                                   f"{npairs_total} unordered pairs (both directions + mutual check)",
                       "pairs_per_rank": npairs, "mutual_matches_rank0": n_matches,
                       "parallelism": f"pairs sharded over {world} GPU(s)" + (" + RCCL all-gather" if world > 1 else "")},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": I8_MFMA_PEAK_TOPS, "unit": "TFLOP/s",
                         "frac": achieved / I8_MFMA_PEAK_TOPS,
                         # HBM bytes of one full-batch launch (1872 pairs x 8 workgroups x 256 threads)
                         "traffic": measured_traffic("eacham::match_tile_kernel<8, 2>") if args.kpts == 2000 and args.dim == 256 else None,
                         "kernel": "match_tile_kernel<8, 2>", "launches": launches,
                         "avg_launch_ms": tile_ms / max(launches, 1),
                         "finalize_ms_per_step": fin_ms / args.steps},
        }
        if ba_out is not None:
            out["ba"] = ba_out
        if world == 1 and args.cpu_pairs != 0:
            out["cpu_baseline"] = cpu_baseline(descs, pairs_all, args)
            if ba_out is not None:
                out["ba"]["cpu_baseline"] = cpu_baseline_ba(scene)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def measured_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this same command
    (profiles/r01_pmc_hbm_traffic.json, written by tools/pmc_traffic_json.py; FETCH_SIZE and WRITE_SIZE in
    separate passes, KB units, FETCH doubled per the gfx950 correction of MI355X_MICROARCH.md). The entry
    of the largest grid = the full-batch launch the roofline line is about. None if not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")) as f:
            tab = json.load(f)
        keys = [k for k in tab if k.startswith(kernel + " grid=")]
        d = tab[max(keys, key=lambda k: int(k.rsplit("=", 1)[1]))]
        return (2.0 * d["FETCH_SIZE_KB_mean_per_dispatch"] + d["WRITE_SIZE_KB_mean_per_dispatch"]) * 1024.0
    except (OSError, KeyError, TypeError, ValueError):
        return None


def host_cores() -> int:
    """Usable host cores: the affinity mask capped by the cgroup CPU quota (the GPU box grants a
    share of a large host)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def bench_ba(ctx, scene, args, rank, world, dev):
    """BA iters/s: K timed RefineBA solves (config/SfmConfigNerf.json `refine_ba`: LM, 100 it, 1e-5) of
    the whole S200 window, values resident on the device, each solve restarting from the same
    perturbed initial guess. BA does not shard at these sizes: with N > 1 every
    rank runs an independent replica and the rates are summed ("replicas")."""
    import torch
    import torch.distributed as dist
    from eacham_amd import ba, capi

    arrays = ba.BaArrays.from_scene(scene)
    cfg = ba.OptimizerConfig.refine_ba()
    solver = ba.PreparedBA(ctx, arrays)
    first = solver.run(cfg)  # warm-up (allocations, code objects)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outer = inner = 0
    for _ in range(args.ba_solves):  # the rate: no instrumentation inside the timed region
        o = solver.run(cfg)
        outer += o.outer_iterations
        inner += o.inner_iterations
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the per-stage breakdown comes from one more solve with the event timers on (they cost ~7 % between
    # the small kernels, so they stay out of the rate)
    ctx.profile_reset()
    ctx.profile_enable(True)
    prof = solver.run(cfg)
    torch.cuda.synchronize()
    ctx.profile_enable(False)
    stage = {}
    for name, kid in [("linearize", capi.KERNEL_BA_LINEARIZE), ("schur", capi.KERNEL_BA_SCHUR),
                      ("solve", capi.KERNEL_BA_SOLVE), ("error", capi.KERNEL_BA_ERROR)]:
        n, ms = ctx.profile_get(kid)
        stage[name + "_ms_per_inner_iter"] = ms / max(prof.inner_iterations, 1)
    solver.close()
    rate = torch.tensor([outer / dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(rate, op=dist.ReduceOp.SUM)
    nc, nl, no = arrays.cam_T_wc.shape[0], arrays.points.shape[0], arrays.obs_cam.shape[0]
    n = 6 * nc + 5
    bytes_iter = 48 * no + 3 * (96 * nc + 24 * nl + 40) + 144 * nl + 8 * n * (n + 1)
    dev_ms = sum(stage.values())
    achieved = bytes_iter / (dev_ms * 1e-3) / 1e9 if dev_ms > 0 else 0.0
    return {"value": float(rate.item()), "unit": "LM outer iters/s", "replicas": world, "solves": args.ba_solves,
            "outer_iters_per_solve": outer / args.ba_solves, "inner_iters_per_solve": inner / args.ba_solves,
            "ms_per_inner_iter": dt / max(inner, 1) * 1e3, "dtype": "f64",
            "workload": f"S200 RefineBA: {nc} cams / {nl} landmarks / {no} obs, refine_ba (LM, 100, 1e-5)",
            "final_error": first.final_error, "initial_error": first.initial_error, **stage,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": None,
                         "note": "algorithmic bytes of one inner iteration (SURVEY.md §8(d)) / summed kernel time",
                         # the dense reduced solve against the fp64 vector peak (SURVEY.md §8(d) asks for both)
                         "solve": {"bound": "fp64", "achieved": (n ** 3 / 3.0) / (stage["solve_ms_per_inner_iter"] * 1e-3) / 1e12
                                   if stage["solve_ms_per_inner_iter"] > 0 else 0.0,
                                   "peak": 78.6, "unit": "TFLOP/s",
                                   "frac": ((n ** 3 / 3.0) / (stage["solve_ms_per_inner_iter"] * 1e-3) / 1e12 / 78.6)
                                   if stage["solve_ms_per_inner_iter"] > 0 else 0.0,
                                   "note": "n^3/3 flops of the Cholesky factorisation / solve time; the chain of 38 dependent block steps is latency-bound"}}}


def cpu_baseline_ba(scene):
    """oracle/ba_oracle.c (kind "port"): two LM iterations of the same S200 window on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api as O
    from eacham_amd import ba
    cores = host_cores()
    arrays = ba.BaArrays.from_scene(scene)
    O.ba_solve(arrays, ba.OptimizerConfig("LM", 1, 1e-5, 10.0, False), nthreads=cores)  # warm-up
    t0 = time.perf_counter()
    out = O.ba_solve(arrays, ba.OptimizerConfig("LM", 3, 1e-5, 10.0, False), nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": out.outer_iterations / dt, "unit": "LM outer iters/s", "cores": cores, "kind": "port",
            "sample": f"{out.outer_iterations} LM iterations of the same window (Schur + dense Cholesky, OpenMP), {dt:.1f} s"}


def cpu_baseline(descs, pairs_all, args):
    """The CPU restatement (oracle/match_oracle.c, kind "port") on a bounded sample of the same
    workload, threaded over pairs like apps/sfm/main.cpp:98, on the GPU box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api as O
    cores = host_cores()
    n = args.cpu_pairs if args.cpu_pairs > 0 else 400 * cores  # ~10 s of CPU work on the GPU box's share
    n = min(n, len(pairs_all))
    sel = pairs_all[np.linspace(0, len(pairs_all) - 1, n).astype(np.int64)]
    O.match_all_pairs(descs, sel[:cores], nthreads=cores)  # warm-up (threads, page faults)
    t0 = time.perf_counter()
    res = O.match_all_pairs(descs, sel, nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "image-pairs/s", "cores": int(res[5]), "kind": "port",
            "sample": f"{n} of {len(pairs_all)} pairs of the same workload, exact brute-force 2-NN + ratio + mutual check, {dt:.1f} s"}


if __name__ == "__main__":
    main()
