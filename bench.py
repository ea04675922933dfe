#!/usr/bin/env python3
"""bench.py — headline benchmark of the eacham hot path on MI355X (contract in the task brief).

Metric (BASELINE.json): image-pairs matched/s (+ BA iters/s) on the 200-frame / 50k-landmark
synthetic scene S200. One "step" = one pass of the matching hot path over ALL 19,900 unordered
frame pairs of S200 (2000 keypoints x 256-D per frame): the operand-swapped int8-MFMA row sweep
(match_sweep_kernel: per query row the two smallest distances + the tile of the minimum), ratio
test, the column direction for the candidates' columns only, mutual cross-check, CSR compaction —
and, for N > 1, the RCCL all-gather of the match graph. Inputs are resident in HBM before the
timed region. Every number comes with its parity gate: `parity` = a seeded sample of the LAST
timed step's match graph (incl. the pairs at every launch boundary) against the CPU oracle, bit
for bit; `ba.parity` = the device's solve against the oracle's (1e-5 relative).

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Output: rank 0 prints one JSON line per sub-line as it finishes ({"line": "<name>", ...}: the other BASELINE.json
configurations and the real descriptor shapes, each with its own roofline against the peak of the arithmetic it runs
on, SURVEY.md §8(d) "Other configs as inputs"; {"line": "ba", ...} is the full bundle-adjustment record) and, LAST,
the headline line of the contract — kept short (headline + roofline + cpu_baseline + a compact `ba` + one
value / roofline fraction per sub-line) so that a tail of the log holds it whole. The sub-lines:
  s200_d128_i8   S200 with 128-D integer descriptors (what eacham's SIFT extractor produces)
  s200_d256_f32  S200 with 256-D unit-norm float descriptors (SuperPoint / LightGlue style), fp32 MFMA
  c2             configs[1]: 100 frames x 2000 x 256-D
  c3_tum         configs[2] stand-in: 500 frames x 600 x 128-D all pairs + a sequence of local-window RefineBA
  c3_sfm_loop    the whole incremental loop of apps/sfm/main.cpp:76-240 on a 100-frame TUM-sized sequence (compiles and runs a
                 C++ driver as a CHILD process: its own line so that profiling runs can leave it out — tools/prof.sh does)
  c4_ba          configs[3]: 500 cams / 100k landmarks / 1M observations, LM inner loop
  c5_kitti       configs[4] stand-in: 100 frames x 1500 x 128-D through the shard path (+ RCCL all-gather, N > 1)
  c5_kitti_long  the 1000-frame variant of the same: 499 500 pairs, ~12 s including the synthesis
With N > 1 only the headline and c5_kitti run (the lines that shard); `--lines` selects explicitly.
`python bench.py --gpus N` without a launcher environment starts its N ranks itself (child processes, before the parent
touches a GPU); `--single-process` runs N GPUs from ONE process through the C-ABI communicator instead.

The CPU oracle (oracle/) is used here only for the `cpu_baseline` leg (rank 0, N=1, bounded sample).
"""
from __future__ import annotations

import argparse
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

I8_MFMA_PEAK_TOPS = 5000.0  # dense int8 MFMA: 2x the ~2.5 PF bf16 dense rate (MI355X_MICROARCH.md, Matrix cores)
F32_MFMA_PEAK_TFLOPS = 157.3  # f32-input MFMA = the fp32 vector rate (same guide)
HBM_PEAK_GBS = 8000.0
FP64_PEAK_TFLOPS = 78.6
ALL_LINES = ["s200_d128_i8", "s200_d256_f32", "c2", "c3_tum", "c3_sfm_loop", "c4_ba", "c5_kitti", "c5_kitti_long"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--kpts", type=int, default=2000)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--landmarks", type=int, default=50_000)
    ap.add_argument("--cpu-pairs", type=int, default=-1, help="pairs in the CPU-baseline sample (-1 = 400 per core, 0 = skip)")
    ap.add_argument("--ba-solves", type=int, default=50, help="timed RefineBA solves of the S200 window per option set (0 = skip BA)")
    ap.add_argument("--lines", default="auto", help="comma list of sub-lines (see the module docstring), 'all', 'none', or "
                                                    "'auto' = all at N=1, c5_kitti at N>1")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N > 1 on one GPU)")
    ap.add_argument("--all-on-device", type=int, default=-1, help="rehearsal: put every rank on this device index")
    ap.add_argument("--single-process", action="store_true",
                    help="N GPUs from ONE process through the C-ABI communicator (eacham_comm_*: a context + host thread per device, "
                         "RCCL all-gather) instead of one process per GPU; needs no launcher")
    return ap.parse_args()


SOURCE_GROUPS = {"match": ("matcher.hip", "matcher_f32.hip", "context.hip", "context.hpp"),
                 "ba": ("ba.hip", "ba_plan.hpp", "devprim.hpp", "context.hpp"),
                 "solve": ("solve.hip", "score.hip", "context.hpp")}
PROFILE_ROUND = "r05"  # profiles/<round>_pmc_*.json read for roofline.traffic


def kernel_source_sha(group: str | None = None) -> str:
    """Identity of the kernels a profile was taken with: sha256 over the CODE of the library's HIP sources — comments
    and white space are dropped first, so rewording a comment does not disown a profile, changing a token does.
    `group` ("match" / "ba") restricts it to the sources of one path: a change to the bundle adjuster does not disown
    the matcher's counters."""
    import re
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "eacham_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "eacham_amd", "csrc", "*.hpp")))
    if group is not None:
        files = [f for f in files if os.path.basename(f) in SOURCE_GROUPS[group]]
    for fn in files:
        with open(fn, "r", encoding="utf-8", errors="replace") as f:
            src = f.read()
        src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)        # block comments
        src = re.sub(r"(?m)(?<![:\"'])//.*$", " ", src)          # line comments (not the // of a URL inside a string)
        h.update(os.path.basename(fn).encode())
        h.update(" ".join(src.split()).encode())
    return h.hexdigest()[:16]


def measured_traffic(kernel_prefix: str, grid: int | None = None):
    """(HBM bytes per launch, source note) of a kernel from the committed rocprofv3 --pmc passes
    (profiles/r02_pmc_hbm_traffic.json, written by tools/pmc_traffic_json.py: FETCH_SIZE and WRITE_SIZE in separate
    passes, KB units, FETCH doubled per the gfx950 correction of MI355X_MICROARCH.md). The file records the sha of
    the kernel sources it was taken with; a figure from other sources is NOT reported (None + the reason)."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_pmc_hbm_traffic.json")
    try:
        with open(path) as f:
            tab = json.load(f)
    except (OSError, ValueError):
        return None, "no PMC profile committed"
    meta = tab.get("__meta__", {})
    want = kernel_source_sha("match") if "kernel_source_sha_match" in meta else kernel_source_sha()
    have = meta.get("kernel_source_sha_match", meta.get("kernel_source_sha"))
    if have != want:
        return None, f"stale: profile taken with kernel sources {have}, running {want}"
    keys = [k for k in tab if k.startswith(kernel_prefix) and " grid=" in k]
    if grid is not None:
        keys = [k for k in keys if int(k.rsplit("=", 1)[1]) == grid] or keys
    if not keys:
        return None, "kernel not in the PMC profile"
    try:
        d = tab[max(keys, key=lambda k: int(k.rsplit("=", 1)[1]))]
        return (2.0 * d["FETCH_SIZE_KB_mean_per_dispatch"] + d["WRITE_SIZE_KB_mean_per_dispatch"]) * 1024.0, \
            f"profiles/{PROFILE_ROUND}_pmc_hbm_traffic.json ({meta.get('command', 'tools/prof.sh')})"
    except (KeyError, TypeError, ValueError):
        return None, "malformed PMC profile entry"


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


class Dist:
    """rank/world + the torch handles every leg needs."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}: launch with torch.distributed.run")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
        if args.all_on_device >= 0:
            local = args.all_on_device
        self.local = local
        torch.cuda.set_device(local)
        self.dev = torch.device("cuda", local)
        self.backend = args.backend
        self.group_note = None
        if self.world > 1:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(args.backend)
        elif args.backend == "nccl":
            # N = 1: a one-rank RCCL group, so that the device-tensor all-gather of the shard path and its stream
            # join execute on every single-GPU run too (the c5_kitti line gathers through it)
            try:
                import socket
                with socket.socket() as so:
                    so.bind(("127.0.0.1", 0))
                    port = so.getsockname()[1]
                dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=self.dev)
            except Exception as e:  # noqa: BLE001 - reported in the line, never fatal for the single-GPU numbers
                self.group_note = f"one-rank nccl group not available: {type(e).__name__}: {e}"

    @property
    def has_group(self) -> bool:
        return self.dist.is_available() and self.dist.is_initialized()

    @property
    def collective_name(self) -> str:
        return "RCCL (torch.distributed backend nccl)" if self.backend == "nccl" else f"torch.distributed backend {self.backend}"

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_f(self, x: float) -> float:
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev)
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_f(self, x: float) -> float:
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev)
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())


def run_matching(D: Dist, descs, kind: str, steps: int, warmup: int, gather_at_one: bool = False):
    """The matching hot path over all unordered pairs of `descs` (list of N x dim fp32 matrices), sharded over
    the ranks: pairs ordered by train frame, contiguous shard per rank, one eacham_match_all_pairs_dev per step
    and (N > 1) the asynchronous RCCL all-gather of the match graph, double-buffered against the next step.
    kind: "i8" (integer descriptors, exact int8 MFMA path) or "f32" (float descriptors, fp32 MFMA path).
    gather_at_one: at N = 1 run the all-gather all the same, through the one-rank group (and check what it returns).
    Returns the timing of exactly `steps` steps bracketed by barrier + synchronize, max over ranks."""
    torch, dist = D.torch, D.dist
    from eacham_amd import HipContext, synth, capi, shard
    world, rank, dev = D.world, D.rank, D.dev
    gather = world > 1 or (gather_at_one and D.has_group)
    pairs_all = shard.order_pairs(synth.all_pairs(len(descs)))
    npairs_total = len(pairs_all)
    # contiguous shards of equal WORK (a pair costs rows(f1) * rows(f2)): the same cut as eacham_shard_bounds_weighted
    bounds = shard.shard_bounds_weighted(shard.pair_weights(pairs_all, [d.shape[0] for d in descs]), world)
    pairs = shard.shard_pairs(pairs_all, world, rank, bounds)
    npairs = len(pairs)
    shard_max = max(int(np.diff(bounds).max()), 1)
    kmax = max(d.shape[0] for d in descs)

    ctx = HipContext(D.local)
    t_up = time.perf_counter()
    for f, d in enumerate(descs):  # replicated descriptor store (S200: 200 x 2000 x 256 B = 102 MB int8)
        (ctx.upload_descriptors if kind == "i8" else ctx.upload_descriptors_f32)(f, d)
    ctx.sync()
    t_up = time.perf_counter() - t_up  # once per job, outside the timed steps: PCIe copy of the fp32 matrices + layout kernels
    ext = torch.cuda.ExternalStream(ctx.stream, device=dev)
    with torch.cuda.stream(ext):
        pairs_dev = torch.from_numpy(pairs).to(dev)
        offsets = torch.zeros(npairs + 1, dtype=torch.int64, device=dev)
        total = torch.zeros(1, dtype=torch.int64, device=dev)
        # size the edge buffer from one untimed pass (deterministic inputs -> exact)
        probe_cap = max(npairs * kmax, 1)
        counts0 = torch.zeros(shard_max, dtype=torch.int32, device=dev)
        edges0 = torch.zeros(probe_cap * 2, dtype=torch.int32, device=dev)
        ctx.match_all_pairs_dev(pairs_dev.data_ptr(), npairs, counts0.data_ptr(), offsets.data_ptr(),
                                edges0.data_ptr(), probe_cap, total.data_ptr())
        ctx.sync()
        cap_t = torch.tensor([int(total.item())], dtype=torch.int64, device=dev)
        if world > 1:
            dist.all_reduce(cap_t, op=dist.ReduceOp.MAX)
        edge_cap = max(int(cap_t.item()), 1)
        del edges0, counts0
        # two output sets: the all-gather of step i (its own RCCL stream) overlaps the matching of step i+1
        sets = []
        for _ in range(2 if gather else 1):
            st = {"counts": torch.zeros(shard_max, dtype=torch.int32, device=dev),
                  "edges": torch.zeros(edge_cap * 2, dtype=torch.int32, device=dev), "pending": []}
            if gather:
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
            if gather:  # all-gather of the match graph (counts + padded edge lists): RCCL over xGMI with the nccl backend
                _, _, st["pending"] = shard.all_gather_match_graph(st["counts"], st["edges"], shard_max, edge_cap, world,
                                                                   st["g_counts"], st["g_edges"], async_op=True)

    def fence():
        with torch.cuda.stream(ext):
            for st in sets:
                for w in st["pending"]:
                    w.wait()
                st["pending"] = []
        D.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    fence()
    ctx.profile_reset()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = D.max_f(time.perf_counter() - t0)
    ctx.profile_enable(False)
    batches = ctx.match_batches(npairs, stats=False) if kind == "i8" else (np.zeros(1, np.int32), 1)
    stream2 = ctx.stream2_info()
    launches, tile_ms = ctx.profile_get(capi.KERNEL_MATCH_TILE)
    _, fin_ms = ctx.profile_get(capi.KERNEL_MATCH_FINALIZE)
    gathered_ok = None
    if gather:  # the last step's gathered graph against this rank's own shard
        st = sets[(step_no[0] - 1) % len(sets)]
        mine_c = st["g_counts"][rank * shard_max:(rank + 1) * shard_max]
        mine_e = st["g_edges"][rank * edge_cap * 2:(rank + 1) * edge_cap * 2]
        gathered_ok = bool(torch.equal(mine_c, st["counts"]) and torch.equal(mine_e, st["edges"]))
    st = sets[(step_no[0] - 1) % len(sets)]  # the LAST timed step's match graph of this rank's shard, for the parity gate
    graph = {"pairs": pairs, "counts": st["counts"][:npairs].cpu().numpy(), "offsets": offsets.cpu().numpy(),
             "edges": st["edges"].cpu().numpy().view(np.uint32).reshape(-1, 2), "force_f32": 0 if kind == "i8" else 2}
    out = {"elapsed": elapsed, "npairs_total": npairs_total, "npairs": npairs, "launches": launches, "tile_ms": tile_ms, "graph": graph, "batch_starts": batches[0], "slots": batches[1], "stream2": stream2,
           "gathered": gather, "gathered_ok": gathered_ok,
           "fin_ms": fin_ms, "matches": int(total.item()), "pairs_all": pairs_all, "edge_cap": edge_cap, "upload_s": t_up,
           "upload_bytes": int(sum(d.nbytes for d in descs))}
    ctx.close()
    return out


def parity_matching(descs, graph, idx, want=None):
    """The parity gate printed with every matching number: pairs `idx` (indices into this rank's shard) of the
    match graph the LAST timed step left on the device against the CPU oracle (oracle/match_oracle.c) — counts and every (q, t),
    bit for bit. `want` = an oracle result for exactly those pairs that somebody already computed (the cpu_baseline leg)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api as O
    idx = np.asarray(idx, dtype=np.int64)
    if want is None:
        want = O.match_all_pairs(descs, graph["pairs"][idx], nthreads=host_cores(), force_f32=graph["force_f32"])
    c, o, e = graph["counts"], graph["offsets"], graph["edges"]
    ok = bool(np.array_equal(c[idx], want[0]))
    got_q = np.concatenate([e[o[p]:o[p] + c[p], 0] for p in idx]) if len(idx) else np.zeros(0, np.uint32)
    got_t = np.concatenate([e[o[p]:o[p] + c[p], 1] for p in idx]) if len(idx) else np.zeros(0, np.uint32)
    ok = ok and bool(np.array_equal(got_q, want[2]) and np.array_equal(got_t, want[3]))
    return {"pairs_checked": int(len(idx)), "edges_checked": int((want[0] > 0).sum()), "matches_checked": int(want[0].sum()),
            "bit_exact": ok, "against": "oracle/match_oracle.c on the last timed step's match graph"}


def parity_sample(ctx_batches, npairs, n, seed):
    """Seeded pair indices incl. both sides of every launch boundary of the job (eacham_match_debug_batches)."""
    rng = np.random.default_rng(seed)
    pick = {0, npairs - 1}
    for s0 in ctx_batches:
        pick.update(int(s0) + d for d in (-1, 0))
    pick.update(rng.choice(npairs, min(n, npairs), replace=False).tolist())
    return np.array(sorted(p for p in pick if 0 <= p < npairs), dtype=np.int64)


def sweep_kernel(dim: int) -> str:
    """The distance sweep run_match launches for `dim`-D integer descriptors without per-pair statistics (matcher.hip: run_match)."""
    ks = 2 if dim <= 64 else 4 if dim <= 128 else 8
    if os.environ.get("EACHAM_MATCH_TILE_SWEEP", "0") not in ("", "0"):
        return f"match_tile_kernel<{ks}, 2, false>"
    # the bound form of the sweep (+ an exact pass over the rows it leaves open) up to 128-D, the exact form at 256-D (matcher.hip: run_match)
    form = os.environ.get("EACHAM_MATCH_SWEEP_FORM", "")
    bound = form == "bound" or (form != "exact" and ks <= 4)
    return f"match_sweep_kernel<{ks}, {'true' if bound else 'false'}>"


def matching_line(D: Dist, descs, kind: str, dim: int, steps: int, warmup: int, workload: str, kernel: str, gather_at_one: bool = False,
                  parity_pairs: int = 48):
    """One matching sub-line: value + roofline against the MFMA peak of the arithmetic used + the parity gate (rank 0: a seeded
    sample of the last step's graph incl. the pairs at every launch boundary against the CPU oracle; parity_pairs = 0 leaves it
    to the caller, who has an oracle result already)."""
    r = run_matching(D, descs, kind, steps, warmup, gather_at_one)
    parity = None
    if D.rank == 0 and parity_pairs > 0 and r["npairs"] > 0:
        parity = parity_matching(descs, r["graph"], parity_sample(r["batch_starts"], r["npairs"], parity_pairs, len(descs) + dim))
    n = np.array([d.shape[0] for d in descs], dtype=np.float64)
    # algorithmic work of this rank's launches: 2 * N1 * N2 * D per unordered pair
    ops = 2.0 * dim * float(np.mean(n)) ** 2 * r["npairs"] * steps
    achieved = ops / (r["tile_ms"] * 1e-3) / 1e12 if r["tile_ms"] > 0 else 0.0
    peak = I8_MFMA_PEAK_TOPS if kind == "i8" else F32_MFMA_PEAK_TFLOPS
    return {"workload": workload, "value": r["npairs_total"] * steps / r["elapsed"], "unit": "image-pairs/s",
            "dtype": kind, "steps": steps, "ms_per_step": r["elapsed"] / steps * 1e3, "pairs": r["npairs_total"],
            "pairs_per_rank": r["npairs"], "mutual_matches_rank0": r["matches"],
            "launches_per_step": int(len(r["batch_starts"])), "workspace_slots": int(r["slots"]),
            "second_stream": r["stream2"],  # which candidate the context's search kept and whether it has a hardware queue of its own
            **({"parity": parity} if parity is not None else {}),
            **({"all_gather": {"collective": D.collective_name, "world": D.world, "gathered_equals_local_shard": r["gathered_ok"]}} if r["gathered"] else {}),
            # the descriptor hand-over happens once per job, before the timed steps (host fp32 -> HBM int8 / fp32 fragments)
            "upload_once": {"seconds": r["upload_s"], "host_bytes": r["upload_bytes"],
                            "pairs_per_s_if_paid_every_step": r["npairs_total"] / (r["elapsed"] / steps + r["upload_s"])},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": None, "kernel": kernel, "launches": r["launches"],
                         "avg_launch_ms": r["tile_ms"] / max(r["launches"], 1),
                         "finalize_ms_per_step": r["fin_ms"] / steps,
                         # the same work over the WHOLE step (sweeps + what runs beside and behind them): the launch time above
                         # includes the candidate-pass kernels that share the chip with a sweep on the second stream
                         "frac_of_step": (ops / r["elapsed"] / 1e12 / peak) if r["elapsed"] > 0 else 0.0}}, r


def ba_bytes_flops(nc, nl, no):
    n = 6 * nc + 5
    return 48 * no + 3 * (96 * nc + 24 * nl + 40) + 144 * nl + 8 * n * (n + 1), n


def bench_ba(D: Dist, ctx, scene, solves: int, cfg, label: str, with_traffic=True):
    """BA iters/s: `solves` timed RefineBA solves of one window, values resident on the device, each solve restarting
    from the same perturbed initial guess. BA does not shard at these sizes: with N > 1 every rank
    runs an independent replica and the rates are summed ("replicas")."""
    torch = D.torch
    from eacham_amd import ba, capi

    arrays = ba.BaArrays.from_scene(scene)
    ba.PreparedBA(ctx, arrays).close()  # the first construction of a size allocates the context's scratch and arenas: not what is reported
    solver = ba.PreparedBA(ctx, arrays)
    plan = solver.plan_info()
    first = solver.run(cfg)  # warm-up (allocations, code objects)
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outer = inner = 0
    for _ in range(solves):  # the rate: no instrumentation inside the timed region
        o = solver.run(cfg, trace_cap=0)
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
        _, ms = ctx.profile_get(kid)
        stage[name + "_ms_per_inner_iter"] = ms / max(prof.inner_iterations, 1)
    solver.close()
    # The CALL as the app issues it (apps/sfm/main.cpp:207,230): eacham_ba_solve = graph-to-structure construction
    # (BundleAdjuster.cpp:57-178: upload + device sorts / scans + the host's elimination plan) + LM + read-back of poses,
    # points and K, a new problem every time. This is the number a caller of RefineBA sees; the loop above is its LM part.
    n_calls = max(3, min(solves, 20))
    ba.RefineBA(ctx, arrays, cfg, trace_cap=0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    outer_c = 0
    for _ in range(n_calls):
        outer_c += ba.RefineBA(ctx, arrays, cfg, trace_cap=0).outer_iterations
    torch.cuda.synchronize()
    dt_call = time.perf_counter() - t1
    rate = D.sum_f(outer / dt)
    nc, nl, no = arrays.cam_T_wc.shape[0], arrays.points.shape[0], arrays.obs_cam.shape[0]
    bytes_iter, n = ba_bytes_flops(nc, nl, no)
    dev_ms = sum(stage.values())
    achieved = bytes_iter / (dev_ms * 1e-3) / 1e9 if dev_ms > 0 else 0.0
    solve_ms = stage["solve_ms_per_inner_iter"]
    # flops the sparse plan EXECUTES: rank-64 updates of 64 x 64 tiles (2 * 64^3 each); the dense n^3/3 figure of SURVEY.md
    # section 8(d) is kept beside it under its own key
    sparse_flops = plan["tile_updates"] * 2.0 * 64 ** 3
    solve_tf = sparse_flops / (solve_ms * 1e-3) / 1e12 if solve_ms > 0 else 0.0
    dense_tf = (n ** 3 / 3.0) / (solve_ms * 1e-3) / 1e12 if solve_ms > 0 else 0.0
    traffic, src = (None, "not collected for this window")
    if with_traffic:
        traffic, src = ba_measured_traffic()
    return {"_outcome": first, "value": rate, "unit": "LM outer iters/s", "replicas": D.world, "solves": solves,
            "outer_iters_per_solve": outer / solves, "inner_iters_per_solve": inner / solves,
            "inner_iters_per_s": D.sum_f(inner / dt), "timed_region_s": dt,
            "ms_per_inner_iter": dt / max(inner, 1) * 1e3, "dtype": "f64",
            "ms_per_solve": dt / solves * 1e3,
            "ms_per_solve_incl_prepare": dt_call / n_calls * 1e3,
            "iters_per_s_incl_prepare": D.sum_f(outer_c / dt_call),
            "calls_incl_prepare": n_calls,
            "workload": f"{label}: {nc} cams / {nl} landmarks / {no} obs, {cfg.method} ({cfg.maxIter}, {cfg.maxTolerance:g})",
            "final_error": first.final_error, "initial_error": first.initial_error, **stage,
            # the analysis of the reduced camera system (ba_plan.hpp): ordering, 64-column panels, tiles of the symbolic
            # factor, height of the elimination tree = dependent factorisation launches per solve
            "plan": plan,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
                         "algorithmic_bytes_per_inner_iter": bytes_iter,
                         "note": "algorithmic bytes of one inner iteration (SURVEY.md §8(d)) / summed kernel time",
                         # the dense reduced solve against the fp64 vector peak (SURVEY.md §8(d) asks for both)
                         "solve": {"bound": "fp64", "achieved": solve_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": solve_tf / FP64_PEAK_TFLOPS, "executed_flops": sparse_flops,
                                   "dense_equivalent": {"flops": n ** 3 / 3.0, "achieved": dense_tf, "frac": dense_tf / FP64_PEAK_TFLOPS},
                                   "note": f"{plan['tile_updates']} rank-64 tile updates (2 * 64^3 flops each) of the sparse factorisation in "
                                           f"{plan['levels']} dependent launches / solve time: latency-bound by the chain of diagonal-tile "
                                           "factors; dense_equivalent = the n^3/3 of a dense Cholesky over the same time"}}}


def ba_measured_traffic():
    """HBM bytes of ONE LM inner iteration of the S200 window: sum over the BA kernels of (2 FETCH + WRITE) per
    dispatch x dispatches per inner iteration, from profiles/<round>_pmc_ba_traffic.json (tools/pmc_traffic_json.py)."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_pmc_ba_traffic.json")
    try:
        with open(path) as f:
            tab = json.load(f)
    except (OSError, ValueError):
        return None, "no PMC profile committed"
    meta = tab.get("__meta__", {})
    want = kernel_source_sha("ba") if "kernel_source_sha_ba" in meta else kernel_source_sha()
    have = meta.get("kernel_source_sha_ba", meta.get("kernel_source_sha"))
    if have != want:
        return None, f"stale: profile taken with kernel sources {have}, running {want}"
    try:
        return float(tab["per_inner_iteration"]["hbm_bytes"]), f"profiles/{PROFILE_ROUND}_pmc_ba_traffic.json"
    except (KeyError, TypeError, ValueError):
        return None, "malformed PMC profile"


def bench_local_windows(D: Dist, ctx, scene, frames, cfg):
    """configs[2] stand-in, BA part: a sequence of local-window RefineBA calls as apps/sfm/main.cpp:207 issues them —
    every window is a NEW problem (eacham_ba_solve = structure build + upload + LM + read-back), so the timed region
    holds all of it."""
    torch = D.torch
    from eacham_amd import ba, synth
    wins = [ba.BaArrays.from_scene(synth.local_window(scene, f)) for f in frames]
    ba.RefineBA(ctx, wins[0], cfg)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outer = inner = 0
    for A in wins:
        o = ba.RefineBA(ctx, A, cfg, trace_cap=0)
        outer += o.outer_iterations
        inner += o.inner_iterations
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cams = np.mean([A.cam_T_wc.shape[0] for A in wins])
    lms = np.mean([A.points.shape[0] for A in wins])
    obs = np.mean([A.obs_cam.shape[0] for A in wins])
    return {"value": outer / dt, "unit": "LM outer iters/s", "windows": len(wins), "windows_per_s": len(wins) / dt,
            "inner_iters_per_s": inner / dt, "timed_region_s": dt, "dtype": "f64",
            "workload": f"{len(wins)} local windows (current frame + covisible neighbours): mean {cams:.0f} cams / {lms:.0f} "
                        f"landmarks / {obs:.0f} obs, refine_ba (LM, 100, 1e-5), prepare + solve + read-back per window"}


def cpu_baseline_ba(scene):
    """oracle/ba_oracle.c (kind "port"): LM iterations of the same S200 window on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api as O
    from eacham_amd import ba
    cores = host_cores()
    arrays = ba.BaArrays.from_scene(scene)
    O.ba_solve(arrays, ba.OptimizerConfig("LM", 1, 1e-5, 10.0, False), nthreads=cores)  # warm-up
    t0 = time.perf_counter()
    iters = solves = 0
    while time.perf_counter() - t0 < 8.0:  # whole RefineBA solves of the same window (refine_ba options) for >= 8 s
        out = O.ba_solve(arrays, ba.OptimizerConfig.refine_ba(), nthreads=cores)
        iters += out.outer_iterations
        solves += 1
    dt = time.perf_counter() - t0
    return {"value": iters / dt, "unit": "LM outer iters/s", "cores": cores, "kind": "port",
            "sample": f"{solves} whole RefineBA calls (each builds its own observation lists, as the device call does) = {iters} LM "
                      f"iterations of the same window (Schur + dense Cholesky, OpenMP), {dt:.1f} s"}, out


def parity_ba(got, ref):
    """The parity gate of the BA numbers: the device's solve of the benchmarked window against the oracle's solve of the same
    window (oracle/ba_oracle.c, the run the cpu_baseline leg makes anyway): same LM decisions, poses / points / K within the
    north-star tolerance of 1e-5 relative."""
    def rel(a, b):
        return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))
    r = {"rel_poses": rel(got.cam_T_wc, ref.cam_T_wc), "rel_points": rel(got.points, ref.points), "rel_K": rel(got.K, ref.K),
         "rel_final_error": abs(got.final_error - ref.final_error) / max(abs(ref.final_error), 1e-300),
         "same_iterations": (got.outer_iterations, got.inner_iterations) == (ref.outer_iterations, ref.inner_iterations),
         "same_decisions": bool(got.trace.shape == ref.trace.shape and np.array_equal(got.trace[:, 3:], ref.trace[:, 3:])),
         "tolerance": 1e-5, "against": "oracle/ba_oracle.c on the same window and options"}
    r["within_tolerance"] = bool(r["same_iterations"] and r["same_decisions"] and max(r["rel_poses"], r["rel_points"], r["rel_K"]) < 1e-5)
    return r


def cpu_baseline(descs, pairs_all, args):
    """The CPU restatement (oracle/match_oracle.c, kind "port") on a bounded sample of the same
    workload, threaded over pairs like apps/sfm/main.cpp:98, on the GPU box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api as O
    cores = host_cores()
    n = args.cpu_pairs if args.cpu_pairs > 0 else 400 * cores  # ~10 s of CPU work on the GPU box's share
    n = min(n, len(pairs_all))
    idx = np.linspace(0, len(pairs_all) - 1, n).astype(np.int64)
    sel = pairs_all[idx]
    O.match_all_pairs(descs, sel[:cores], nthreads=cores)  # warm-up (threads, page faults)
    t0 = time.perf_counter()
    res = O.match_all_pairs(descs, sel, nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "image-pairs/s", "cores": int(res[5]), "kind": "port",
            "sample": f"{n} of {len(pairs_all)} pairs of the same workload, exact brute-force 2-NN + ratio + mutual check, {dt:.1f} s"}, idx, res


def self_launch(args):
    """`python bench.py --gpus N` with no launcher environment: start the N ranks as CHILD processes (the driver's own
    command line: torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) BEFORE this process has touched the
    GPU, and leave with their exit code. Nothing is re-executed in a process that holds a device."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def main_single_process(args):
    """N GPUs from one process: the reference app IS one process whose pair loop fans out over host threads
    (apps/sfm/main.cpp:31, :98-109). eacham_comm_match_run orders the pairs, cuts them by WORK (rows(f1) * rows(f2)), matches
    every shard on its device from its own host thread and assembles the graph on every device with two ncclAllGather calls;
    a step is one such call, the graph left resident on the devices. BA: one independent replica per device, from N host
    threads (replicas only). Prints the contract's JSON line."""
    import threading
    from eacham_amd import HipContext, synth, ba, shard
    N = args.gpus
    scene = synth.make_scene(args.frames, args.landmarks, 10)
    descs, _ = synth.make_frame_descriptors(scene, args.kpts, args.dim)
    pairs = synth.all_pairs(args.frames)
    comm = shard.Comm(N)
    t_up = time.perf_counter()
    for f, d in enumerate(descs):
        comm.upload_descriptors(f, d)
    t_up = time.perf_counter() - t_up
    ctxs = [HipContext.borrowed(comm.ctx_handle(r)) for r in range(N)]
    total = 0
    for _ in range(max(args.warmup, 1)):
        total = comm.match_run(pairs)
    for c in ctxs:
        c.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        total = comm.match_run(pairs)
    for c in ctxs:
        c.sync()
    elapsed = time.perf_counter() - t0
    counts, offsets, q, t = comm.match_fetch()   # outside the timed region: the host copy of device 0's graph
    n = np.array([d.shape[0] for d in descs], dtype=np.float64)
    ops = 2.0 * args.dim * float(np.mean(n)) ** 2 * len(pairs) * args.steps
    achieved = ops / elapsed / 1e12
    out = {"metric": "image-pairs matched/s + BA iters/s, 200-frame/50k-landmark synthetic",
           "value": len(pairs) * args.steps / elapsed, "unit": "image-pairs/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "i8",
           "data": "synthetic",
           "config": {"workload": f"S200 matching: {args.frames} frames x {args.kpts} kpts x {args.dim}-D, {len(pairs)} unordered pairs "
                                  "(both directions + mutual check)",
                      "parallelism": f"single process: {N} device context(s) + host thread(s), work-balanced shards, RCCL ncclAllGather "
                                     "of the match graph (eacham_comm_match_run)"},
           "all_gather": {"collective": "RCCL ncclAllGather (C-ABI communicator)", "world": N, "matches_gathered": int(total),
                          "matches_fetched": int(len(q)), "edges_with_matches": int((counts > 0).sum())},
           "roofline": {"bound": "mfma", "achieved": achieved, "peak": I8_MFMA_PEAK_TOPS * N, "unit": "TFLOP/s",
                        "frac": achieved / (I8_MFMA_PEAK_TOPS * N), "traffic": None,
                        "note": "whole step (matching + all-gather + the host's share) against the int8 peak of the N devices"},
           "upload_once": {"seconds": t_up}, "kernel_source_sha": kernel_source_sha()}
    if args.ba_solves > 0:  # one replica per device, each from its own host thread (ctypes releases the GIL)
        arrays = ba.BaArrays.from_scene(scene)
        cfg = ba.OptimizerConfig.refine_ba()
        rates = [0.0] * N

        def replica(r):
            solver = ba.PreparedBA(ctxs[r], ba.BaArrays.from_scene(scene))
            solver.run(cfg)
            t1 = time.perf_counter()
            outer = sum(solver.run(cfg, trace_cap=0).outer_iterations for _ in range(args.ba_solves))
            rates[r] = outer / (time.perf_counter() - t1)
            solver.close()
        th = [threading.Thread(target=replica, args=(r,)) for r in range(N)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        out["ba"] = {"value": float(sum(rates)), "unit": "LM outer iters/s", "replicas": N, "dtype": "f64",
                     "workload": f"S200 RefineBA: {arrays.cam_T_wc.shape[0]} cams / {arrays.points.shape[0]} landmarks / "
                                 f"{arrays.obs_cam.shape[0]} obs, LM (100, 1e-05), one replica per device"}
    print(json.dumps(out), flush=True)
    for c in ctxs:
        c.close()
    comm.close()


def sfm_loop_line():
    # the whole incremental loop of apps/sfm/main.cpp:76-240 (match -> FindBestPair -> per frame PnP / TriangulateFrame /
    # RefineBA / TriangulateFrame -> global BA) through the reference-typed entry points, 100 frames x 600 kpts from
    # keypoints + descriptors alone, held against the scene's ground truth (tests/cpp/sfm_loop_driver.cpp).
    # A child process with its own context: the host compile of the driver is reported apart from the loop's time.
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sfm_loop_rate
    # The child is the first process to use the device on a fresh box and its timed part lasts 0.2-0.3 s: like every other line it
    # gets one untimed warm-up pass (a first run pays the code-object loads and the clock ramp: 335 against 430-450 frames/s); the
    # first run's figure rides along as `cold_frames_per_s`.
    try:
        cold = sfm_loop_rate.run(100, 600, 6000, 10, 4.0)
        loop = sfm_loop_rate.run(100, 600, 6000, 10, 4.0)
    except Exception as e:  # a missing host compiler must not take the bench down: say so in the line
        return {"value": 0.0, "unit": "frames/s", "error": repr(e)[:300], "roofline": {"frac": 0.0}}
    loop["cold_frames_per_s"], loop["cold_sfm_ms"] = cold.get("frames_per_s", 0.0), cold.get("sfm_ms", 0.0)
    loop.pop("driver", None)
    return {"value": loop.get("frames_per_s", 0.0), "unit": "frames/s", "sfm_loop": loop,
            "workload": "apps/sfm/main.cpp:76-240 on 100 frames x 600 kpts x 128-D (TUM-sized), reference-typed entry points",
            "roofline": {"frac": 0.0, "note": "host-driven loop of microsecond kernels: no roofline claim"}}


def main():
    args = parse()
    if args.single_process:
        return main_single_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    env_world, env_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if args.lines == "auto":
        lines = list(ALL_LINES) if env_world == 1 else ["c5_kitti"]
    elif args.lines in ("none", ""):
        lines = []
    elif args.lines == "all":
        lines = list(ALL_LINES)
    else:
        lines = [x for x in args.lines.split(",") if x]
        bad = [x for x in lines if x not in ALL_LINES]
        if bad:
            raise SystemExit(f"unknown --lines {bad}; known: {ALL_LINES}")
    # The incremental loop is a CHILD process with its own device context. It runs before this process touches the GPU: beside a
    # parent that holds contexts of its own, every one of the child's microsecond launches pays for two processes sharing the
    # device (284-298 frames/s inside the bench against 320 alone). Its line is emitted at its place further down.
    early_loop = sfm_loop_line() if "c3_sfm_loop" in lines and env_rank == 0 else None
    D = Dist(args)
    from eacham_amd import HipContext, synth, ba

    # ---- headline: S200 matching, identical inputs on every rank ----------------------------
    scene = synth.make_scene(args.frames, args.landmarks, 10)
    descs, _ = synth.make_frame_descriptors(scene, args.kpts, args.dim)
    head, r = matching_line(D, descs, "i8", args.dim, args.steps, args.warmup,
                            f"S200 matching: {args.frames} frames x {args.kpts} kpts x {args.dim}-D, "
                            f"{len(synth.all_pairs(args.frames))} unordered pairs (both directions + mutual check)",
                            sweep_kernel(args.dim), parity_pairs=0 if (D.world == 1 and args.cpu_pairs != 0) else 48)
    if args.kpts == 2000 and args.dim == 256:
        head["roofline"]["traffic"], head["roofline"]["traffic_source"] = measured_traffic("eacham::" + sweep_kernel(256))
    else:
        head["roofline"]["traffic_source"] = "not the profiled workload"

    sub = {}
    ctx = HipContext(D.local)  # BA legs
    ba_out = None
    if args.ba_solves > 0:
        ba_out = bench_ba(D, ctx, scene, args.ba_solves, ba.OptimizerConfig.refine_ba(), "S200 RefineBA")
        ba_out["global_ba"] = bench_ba(D, ctx, scene, args.ba_solves, ba.OptimizerConfig.global_ba(), "S200 RefineBA(-1)",
                                       with_traffic=False)

    def emit(name, obj):  # one JSON line per sub-record, as soon as it exists; the contract's line comes last
        if D.rank == 0:
            print(json.dumps({"line": name, **obj}), flush=True)

    ba_first = None
    if ba_out is not None:
        ba_first = ba_out.pop("_outcome")
        ba_out["global_ba"].pop("_outcome", None)
        emit("ba", ba_out)

    def leg(name, fn):
        if name in lines:
            t0 = time.time()
            sub[name] = fn()
            sub[name]["wall_s_incl_setup"] = round(time.time() - t0, 2)
            emit(name, sub[name])

    sub_steps = max(1, min(args.steps, 3))
    leg("s200_d128_i8", lambda: matching_line(
        D, synth.make_frame_descriptors(scene, args.kpts, 128)[0], "i8", 128, sub_steps, 1,
        f"S200 matching with SIFT-shaped descriptors: {args.frames} frames x {args.kpts} kpts x 128-D integers "
        "(FeatureExtractorSift.cpp:8)", sweep_kernel(128))[0])

    def f32_line():
        base = synth.unit_float_descriptors(args.kpts, 256, 1, 99)
        fd = [synth.unit_float_descriptors(args.kpts, 256, 1, f, shared=base[:args.kpts // 2]) for f in range(args.frames)]
        return matching_line(D, fd, "f32", 256, 1, 1,
                             f"S200-sized matching with float descriptors: {args.frames} frames x {args.kpts} kpts x 256-D unit-norm fp32 "
                             "(SuperPoint / LightGlue, modules/onnx/lightglue/feature/Types.h:11-14)", "match_tile_f32_kernel")[0]
    leg("s200_d256_f32", f32_line)
    leg("c2", lambda: matching_line(
        D, descs[:100] if args.frames >= 100 else descs, "i8", args.dim, sub_steps, 1,
        "BASELINE configs[1]: brute-force 256-D descriptor match, 2k kpts x 100 synthetic frames (4950 pairs)",
        sweep_kernel(args.dim))[0])

    def tum_line():
        # config/ConfigTUM.json:3,28: <= 500 frames x 600 features (it asks for ORB/Hamming; the path stays L2 on
        # SIFT-shaped 128-D integers); 30 000 landmarks x 10 observers = 600 per frame
        tum = synth.make_scene(500, 30_000, 10, seed=3)
        td, _ = synth.make_frame_descriptors(tum, 600, 128, seed=3)
        out, _r = matching_line(D, td, "i8", 128, sub_steps, 1,
                                "BASELINE configs[2] stand-in (TUM fr1/desk sizes): 500 frames x 600 kpts x 128-D, 124750 pairs",
                                sweep_kernel(128))
        out["ba"] = bench_local_windows(D, ctx, tum, range(100, 140), ba.OptimizerConfig.refine_ba())
        return out

    leg("c3_tum", tum_line)
    leg("c3_sfm_loop", lambda: early_loop if early_loop is not None else {"value": 0.0, "unit": "frames/s", "roofline": {"frac": 0.0}})
    def c4_line():
        o = bench_ba(D, ctx, synth.make_scene(500, 100_000, 10, seed=4), max(3, args.ba_solves // 10),
                     ba.OptimizerConfig.refine_ba(), "BASELINE configs[3]", with_traffic=False)
        o.pop("_outcome", None)
        return o
    leg("c4_ba", c4_line)

    def kitti_line():
        # config/ConfigKITTI.json:3,29: 100 frames x 1500 features; sharded over the ranks + all-gather
        kit = synth.make_scene(100, 15_000, 10, seed=5)
        kd, _ = synth.make_frame_descriptors(kit, 1500, 128, seed=5)
        out, _r = matching_line(D, kd, "i8", 128, sub_steps, 1,
                                f"BASELINE configs[4] stand-in (KITTI seq-00 sizes): 100 frames x 1500 kpts x 128-D, 4950 pairs sharded "
                                f"over {D.world} GPU(s) + all-gather of the match graph ({D.collective_name})",
                                sweep_kernel(128), gather_at_one=True)
        if D.group_note:
            out["all_gather_note"] = D.group_note
        out["scaling"] = "strong"
        return out
    leg("c5_kitti", kitti_line)

    def kitti_long_line():
        # the longer KITTI-like sequence of SURVEY.md §8(d): 1000 frames x 1500 x 128-D, 499 500 pairs through the shard path
        kit = synth.make_scene(1000, 150_000, 10, seed=6)
        kd, _ = synth.make_frame_descriptors(kit, 1500, 128, seed=6)
        out, _r = matching_line(D, kd, "i8", 128, 1, 1,
                                f"KITTI-like long sequence: 1000 frames x 1500 kpts x 128-D, 499500 pairs sharded over {D.world} GPU(s)"
                                + (f" + all-gather of the match graph ({D.collective_name})" if D.world > 1 else ""), sweep_kernel(128))
        out["scaling"] = "strong"
        return out
    leg("c5_kitti_long", kitti_long_line)

    if D.rank == 0:
        def compact_roofline(rf, keys=("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "frac_of_step")):
            return {k: (round(v, 6) if isinstance(v, float) else v) for k, v in rf.items() if k in keys}

        out = {
            "metric": "image-pairs matched/s + BA iters/s, 200-frame/50k-landmark synthetic",
            "value": head["value"],
            "unit": "image-pairs/s",
            "n_gpus": D.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "i8",
            "data": "synthetic",
            "config": {"workload": head["workload"], "pairs_per_rank": head["pairs_per_rank"],
                       "parallelism": f"pairs sharded over {D.world} GPU(s)" + (f" + all-gather ({D.collective_name})" if D.world > 1 else "")},
            "roofline": compact_roofline(head["roofline"]),
            "kernel_source_sha": kernel_source_sha(),
            "second_stream": head["second_stream"],
        }
        if D.world == 1 and args.cpu_pairs != 0:
            # the CPU-baseline leg's oracle output IS the parity sample of the headline: the same pairs of the last timed step's graph
            out["cpu_baseline"], cb_idx, cb_res = cpu_baseline(descs, r["pairs_all"], args)
            par = parity_matching(descs, r["graph"], cb_idx, want=cb_res)
            edge = parity_matching(descs, r["graph"], parity_sample(r["batch_starts"], r["npairs"], 0, 0))  # both sides of every launch boundary
            out["parity"] = {"pairs_checked": par["pairs_checked"] + edge["pairs_checked"], "matches_checked": par["matches_checked"] + edge["matches_checked"],
                             "bit_exact": par["bit_exact"] and edge["bit_exact"], "against": par["against"],
                             "launch_boundaries": [int(x) for x in r["batch_starts"]], "workspace_slots": int(r["slots"])}
        elif "parity" in head:
            out["parity"] = head["parity"]
        if ba_out is not None:  # compact: the full record is the {"line": "ba"} line above
            g = ba_out["global_ba"]
            out["ba"] = {"value": ba_out["value"], "unit": ba_out["unit"], "ms_per_inner_iter": round(ba_out["ms_per_inner_iter"], 5),
                         "inner_iters_per_s": ba_out["inner_iters_per_s"], "workload": ba_out["workload"], "dtype": "f64",
                         "replicas": ba_out["replicas"], "solve_ms_per_inner_iter": round(ba_out["solve_ms_per_inner_iter"], 5),
                         "roofline": compact_roofline(ba_out["roofline"]), "plan": ba_out["plan"],
                         # the CALL (graph-to-structure construction + LM + read-back, a new problem each time): what a caller of
                         # RefineBA sees; "value" above is the LM loop of a prepared problem
                         "ms_per_solve": round(ba_out["ms_per_solve"], 4),
                         "ms_per_solve_incl_prepare": round(ba_out["ms_per_solve_incl_prepare"], 4),
                         "iters_per_s_incl_prepare": ba_out["iters_per_s_incl_prepare"],
                         "global_ba": {"value": g["value"], "ms_per_inner_iter": round(g["ms_per_inner_iter"], 5), "workload": g["workload"],
                                       "ms_per_solve_incl_prepare": round(g["ms_per_solve_incl_prepare"], 4),
                                       "iters_per_s_incl_prepare": g["iters_per_s_incl_prepare"]}}
            if D.world == 1 and args.cpu_pairs != 0:
                out["ba"]["cpu_baseline"], ref_out = cpu_baseline_ba(scene)
                out["ba"]["parity"] = parity_ba(ba_first, ref_out)
        if sub:  # one value and one roofline fraction per sub-line; the full records are the lines above
            out["lines"] = {}
            for name, v in sub.items():
                c = {"value": v["value"], "unit": v["unit"], "frac": round(v["roofline"]["frac"], 5)}
                if "ms_per_inner_iter" in v:
                    c["ms_per_inner_iter"] = round(v["ms_per_inner_iter"], 5)
                if "ms_per_solve_incl_prepare" in v:
                    c["ms_per_solve_incl_prepare"] = round(v["ms_per_solve_incl_prepare"], 4)
                if "ba" in v:
                    c["ba_windows_per_s"] = v["ba"]["windows_per_s"]
                if "sfm_loop" in v and "frames_per_s" in v["sfm_loop"]:
                    c["sfm_loop_frames_per_s"] = v["sfm_loop"]["frames_per_s"]
                if "all_gather" in v:
                    c["all_gather_ok"] = v["all_gather"]["gathered_equals_local_shard"]
                if "parity" in v:
                    c["parity"] = {k: v["parity"][k] for k in ("pairs_checked", "bit_exact") if k in v["parity"]} \
                        if "pairs_checked" in v["parity"] else v["parity"]
                out["lines"][name] = c
        print(json.dumps(out), flush=True)
    ctx.close()
    if D.world > 1:
        D.dist.barrier()
    if D.has_group:
        D.dist.destroy_process_group()


if __name__ == "__main__":
    main()
