"""Host-side mirror of the reference bundle-adjustment interface, on top of the C-ABI.

  RefineBA(currentFrameId, graph, map, K, config)   modules/sfm/reconstruction/BundleAdjuster.h:13-17
  OptimizerConfig                                    modules/sfm/config/SfmConfig.h:15-22

The reference walks its Graph/Map objects; this driver takes the same information as arrays
(`BaArrays`) — what include/eacham/BundleAdjusterHip.hpp extracts from graph_t / Map.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import capi


@dataclass
class OptimizerConfig:
    """modules/sfm/config/SfmConfig.h:15-22, fields verbatim."""
    method: str = "LM"
    maxIter: int = 100
    maxTolerance: float = 1e-5
    delta: float = 10.0
    usePreconditioner: bool = False

    # config/SfmConfigNerf.json:29-42
    @staticmethod
    def refine_ba():
        return OptimizerConfig("LM", 100, 1e-5, 10.0, False)

    @staticmethod
    def global_ba():
        return OptimizerConfig("LM", 50, 1e-4, 10.0, False)


ORDERINGS = {"auto": capi.BA_ORDER_AUTO, "natural": capi.BA_ORDER_NATURAL, "rcm": capi.BA_ORDER_RCM, "nd": capi.BA_ORDER_ND}


@dataclass
class BaArrays:
    """The BA window as plain arrays (see eacham_ba_problem in include/eacham_hip.h)."""
    cam_T_wc: np.ndarray          # n_cams x 4 x 4 world->camera (Node::transform)
    cam_fixed: np.ndarray         # n_cams int32
    points: np.ndarray            # n_points x 3
    point_observers: np.ndarray   # n_points int32 (global observer counts)
    obs_cam: np.ndarray           # n_obs uint32
    obs_point: np.ndarray         # n_obs uint32
    obs_uv: np.ndarray            # n_obs x 2
    K: np.ndarray                 # fx, fy, cx, cy
    ordering: str = "auto"        # elimination order of the reduced camera system (EACHAM_BA_ORDER_*): auto | natural | rcm | nd
    _keep: list = field(default_factory=list, repr=False)

    @staticmethod
    def from_scene(scene, initial=True):
        """Arrays of a eacham_amd.synth scene (initial guess or ground truth)."""
        return BaArrays(
            cam_T_wc=np.ascontiguousarray(scene["T_init" if initial else "T_true"], dtype=np.float64),
            cam_fixed=np.ascontiguousarray(scene["fixed"], dtype=np.int32),
            points=np.ascontiguousarray(scene["points_init" if initial else "points_true"], dtype=np.float64),
            point_observers=np.ascontiguousarray(scene["observers"], dtype=np.int32),
            obs_cam=np.ascontiguousarray(scene["obs_cam"], dtype=np.uint32),
            obs_point=np.ascontiguousarray(scene["obs_lm"], dtype=np.uint32),
            # keypoints are cv::Point2f in the reference: pixel coordinates carry fp32 precision
            obs_uv=np.ascontiguousarray(scene["obs_uv"].astype(np.float32), dtype=np.float64),
            K=np.ascontiguousarray(scene["K"], dtype=np.float64))

    def c_problem(self) -> capi.BaProblem:
        arrs = [np.ascontiguousarray(self.cam_T_wc, np.float64).reshape(-1, 16),
                np.ascontiguousarray(self.cam_fixed, np.int32),
                np.ascontiguousarray(self.points, np.float64).reshape(-1, 3),
                np.ascontiguousarray(self.point_observers, np.int32),
                np.ascontiguousarray(self.obs_cam, np.uint32),
                np.ascontiguousarray(self.obs_point, np.uint32),
                np.ascontiguousarray(self.obs_uv, np.float64).reshape(-1, 2)]
        self._keep = arrs
        p = capi.BaProblem()
        p.n_cams, p.n_points, p.n_obs = arrs[0].shape[0], arrs[2].shape[0], arrs[4].shape[0]
        p.ordering = ORDERINGS[self.ordering]
        (p.cam_T_wc, p.cam_fixed, p.points, p.point_observers, p.obs_cam, p.obs_point, p.obs_uv) = [a.ctypes.data for a in arrs]
        for i in range(4):
            p.K[i] = float(self.K[i])
        return p


@dataclass
class BaOutcome:
    cam_T_wc: np.ndarray
    points: np.ndarray
    K: np.ndarray
    initial_error: float
    final_error: float
    final_lambda: float
    status: int
    outer_iterations: int
    inner_iterations: int
    trace: np.ndarray  # rows: lambda, new_error, lin_change, accepted, outer
    reserved: int = 0  # eacham_ba_result.reserved (the CPU oracle reports its PCG iteration total here)


LM_FACTOR_POLICIES = {"reset": capi.BA_LM_FACTOR_RESET, "double": capi.BA_LM_FACTOR_DOUBLE}


def c_options(cfg: OptimizerConfig, min_landmarks: int = 50, lm_factor: str = "reset") -> capi.BaOptions:
    """`lm_factor` is not a field of the reference's OptimizerConfig: it names the reading of GTSAM's
    LevenbergMarquardtState::decreaseLambda to follow (EACHAM_BA_LM_FACTOR_* in include/eacham_hip.h)."""
    o = capi.BaOptions()
    o.method = {"LM": capi.BA_LM, "DogLeg": capi.BA_DOGLEG}[cfg.method]
    o.max_iter, o.max_tolerance, o.delta = int(cfg.maxIter), float(cfg.maxTolerance), float(cfg.delta)
    o.use_preconditioner, o.min_landmarks = int(bool(cfg.usePreconditioner)), int(min_landmarks)
    o.lm_factor_policy = LM_FACTOR_POLICIES[lm_factor]
    return o


def run_solver(fn, arrays: BaArrays, cfg: OptimizerConfig, min_landmarks: int = 50, trace_cap: int = 1024, extra=(),
               lm_factor: str = "reset", tweak=None):
    """Shared marshalling for any function with the eacham_ba_solve result contract. `tweak(options)` may edit
    the eacham_ba_options before the call (test hooks)."""
    prob, opt = arrays.c_problem(), c_options(cfg, min_landmarks, lm_factor)
    if tweak is not None:
        tweak(opt)
    T = np.zeros((prob.n_cams, 16), np.float64)
    pts = np.zeros((prob.n_points, 3), np.float64)
    trace = (capi.BaTraceRow * max(trace_cap, 1))()
    res = capi.BaResult()
    res.cam_T_wc, res.points = T.ctypes.data, pts.ctypes.data
    res.trace_cap, res.trace = trace_cap, C.cast(trace, C.POINTER(capi.BaTraceRow))
    rc = fn(C.byref(prob), C.byref(opt), C.byref(res), *extra)
    tr = np.array([[r.lambda_, r.new_error, r.lin_change, r.accepted, r.outer] for r in trace[:res.trace_len]],
                  dtype=np.float64).reshape(-1, 5)
    out = BaOutcome(T.reshape(-1, 4, 4), pts, np.array(list(res.K)), res.initial_error, res.final_error,
                    res.final_lambda, res.status, res.outer_iterations, res.inner_iterations, tr, int(res.reserved))
    return rc, out


def RefineBA(ctx, arrays: BaArrays, config: OptimizerConfig, min_landmarks: int = 50, trace_cap: int = 1024,
             lm_factor: str = "reset") -> BaOutcome:
    """RefineBA on the device (eacham_ba_solve). `ctx` is a HipContext."""
    L = capi.lib()
    rc, out = run_solver(lambda p, o, r: L.eacham_ba_solve(ctx.handle, p, o, r), arrays, config, min_landmarks, trace_cap,
                         lm_factor=lm_factor)
    ctx._check(rc)
    return out


def debug_step(ctx, arrays: BaArrays, lam: float):
    """eacham_ba_debug_step: reduced system, step and errors of one damped step at the initial values."""
    L = capi.lib()
    prob = arrays.c_problem()
    n = 6 * prob.n_cams + 5
    S = np.zeros((n, n)); g = np.zeros(n); dc = np.zeros(n); dl = np.zeros((prob.n_points, 3))
    err = C.c_double(0); lin = C.c_double(0)
    ctx._check(L.eacham_ba_debug_step(ctx.handle, C.byref(prob), lam, S.ctypes.data, g.ctypes.data, dc.ctypes.data,
                                      dl.ctypes.data, C.addressof(err), C.addressof(lin)))
    return S, g, dc, dl, err.value, lin.value


class PreparedBA:
    """eacham_ba_prepare / eacham_ba_run / eacham_ba_release: solve one window repeatedly (benchmark)."""

    def __init__(self, ctx, arrays: BaArrays):
        self.ctx, self.arrays, self._L = ctx, arrays, capi.lib()
        self._prob = arrays.c_problem()
        h = C.c_void_p()
        ctx._check(self._L.eacham_ba_prepare(ctx.handle, C.byref(self._prob), C.byref(h)))
        self._h = h

    def run(self, config: OptimizerConfig, min_landmarks: int = 50, trace_cap: int = 1024, lm_factor: str = "reset") -> BaOutcome:
        rc, out = run_solver(lambda p, o, r: self._L.eacham_ba_run(self.ctx.handle, self._h, o, r), self.arrays,
                             config, min_landmarks, trace_cap, lm_factor=lm_factor)
        self.ctx._check(rc)
        return out

    def plan_info(self) -> dict:
        """eacham_ba_get_plan_info: what the analysis of the reduced camera system decided (panels, tiles, tree height, ordering)."""
        info = capi.BaPlanInfo()
        self.ctx._check(self._L.eacham_ba_get_plan_info(self.ctx.handle, self._h, C.byref(info)))
        return {"panels": info.n_panels, "tiles": info.n_tiles, "levels": info.n_levels,
                "ordering": {v: k for k, v in ORDERINGS.items()}[info.ordering], "nd_leaf": info.nd_leaf,
                "tile_updates": int(info.tile_updates), "est_us": round(info.est_us, 1),
                "prepare_us": [round(v, 1) for v in info.prepare_us]}

    STRUCTURE = {"lm_ptr": (0, np.int32), "cam_ptr": (1, np.int32), "cam_obs": (2, np.int32), "obs_pos": (3, np.int32),
                 "obs_cam": (4, np.uint32), "obs_lm": (5, np.uint32), "obs_uv": (6, np.float64), "cam_uv": (7, np.float64),
                 "cam_lm": (8, np.int32), "pos_cam": (9, np.int32), "cam_chunks": (10, np.int32), "cam_chunk_ptr": (11, np.int32),
                 "blocks": (12, np.int32), "pair_chunks": (13, np.int32), "pair_entries": (14, np.int32), "pose0": (15, np.float64),
                 "pt0": (16, np.float64), "lmprior": (17, np.float64), "K0": (18, np.float64), "fixed": (19, np.int32),
                 # the landmark-major structure of the Schur stage (csrc/ba_groups.hpp); empty when the pair lists serve
                 "g_groups": (20, np.int32), "g_lmid": (21, np.int32), "g_lmrow": (22, np.int32), "g_rowinfo": (23, np.int32),
                 "g_uv": (24, np.float64), "g_ent": (25, np.uint32), "g_chunks": (26, np.int32), "g_laneinfo": (27, np.uint32),
                 "g_blk": (28, np.int32), "g_longblk": (29, np.int32),
                 # the dense form for a local window (csrc/ba_window.hpp); empty when another form serves
                 "w_groups": (30, np.int32), "w_lmid": (31, np.int32), "w_lmrow": (32, np.int32), "w_rowinfo": (33, np.int32),
                 "w_uv": (34, np.float64)}

    def structure(self, name: str) -> np.ndarray:
        """eacham_ba_debug_structure: one array of the device-side structure (tests)."""
        which, dtype = self.STRUCTURE[name]
        n = C.c_int64(0)
        rc = self._L.eacham_ba_debug_structure(self.ctx.handle, self._h, which, None, 0, C.byref(n))
        if rc not in (capi.OK, capi.ERR_CAPACITY):
            self.ctx._check(rc)
        out = np.zeros(n.value // np.dtype(dtype).itemsize, dtype)
        if n.value:
            self.ctx._check(self._L.eacham_ba_debug_structure(self.ctx.handle, self._h, which, out.ctypes.data, n.value, C.byref(n)))
        return out

    def close(self):
        if self._h:
            self._L.eacham_ba_release(self.ctx.handle, self._h)
            self._h = None
