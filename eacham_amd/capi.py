"""ctypes binding of the C-ABI in include/eacham_hip.h (libeacham_hip.so, built by csrc/Makefile).

There is no CPU fallback: if the HIP library is missing or no device is present every entry point
raises. PyTorch is not involved here; device buffers are passed as raw pointers.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EACHAM_HIP_LIB selects another build of the same library (kernel A/B experiments)
LIB_PATH = os.environ.get("EACHAM_HIP_LIB") or os.path.join(_HERE, "lib", "libeacham_hip.so")

OK = 0
ERR_INVALID, ERR_HIP, ERR_CAPACITY, ERR_UNSUPPORTED, ERR_NOT_INTEGER, ERR_NO_DEVICE = -1, -2, -3, -4, -5, -6

KERNEL_MATCH_TILE = 0
KERNEL_MATCH_FINALIZE = 1
KERNEL_BA_LINEARIZE = 2
KERNEL_BA_SCHUR = 3
KERNEL_BA_SOLVE = 4
KERNEL_BA_ERROR = 5
KERNEL_TRIANGULATE = 6
KERNEL_SCORE = 7
SCORE_ESSENTIAL, SCORE_HOMOGRAPHY, SCORE_PNP = 0, 1, 2
SOLVE_HOMOGRAPHY4, SOLVE_ESSENTIAL5 = 0, 1


class EachamError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"eacham_hip error {code}: {msg}")
        self.code = code


_lib = None


def lib() -> C.CDLL:
    """Loads libeacham_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C eacham_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback for the hot path")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
    L.eacham_ctx_create.argtypes = [i32, C.POINTER(vp)]
    L.eacham_ctx_destroy.argtypes = [vp]
    L.eacham_ctx_destroy.restype = None
    L.eacham_last_error.argtypes = [vp]
    L.eacham_last_error.restype = C.c_char_p
    L.eacham_ctx_sync.argtypes = [vp]
    L.eacham_ctx_stream2_info.argtypes = [vp, C.POINTER(i32), C.POINTER(C.c_float)]
    L.eacham_ctx_stream.argtypes = [vp]
    L.eacham_ctx_stream.restype = vp
    L.eacham_version.restype = C.c_char_p
    L.eacham_upload_descriptors.argtypes = [vp, i32, vp, i32, i32]
    L.eacham_upload_descriptors_dev.argtypes = [vp, i32, vp, i32, i32]
    L.eacham_upload_descriptors_f32.argtypes = [vp, i32, vp, i32, i32]
    L.eacham_frame_rows.argtypes = [vp, i32]
    L.eacham_clear_descriptors.argtypes = [vp]
    L.eacham_match_pair.argtypes = [vp, i32, i32, dbl, vp, vp, i32, C.POINTER(i32)]
    L.eacham_match_all_pairs.argtypes = [vp, vp, i32, dbl, i32, i32, vp, vp, vp, vp, i64, C.POINTER(i64), vp]
    L.eacham_match_pairs_directed.argtypes = [vp, vp, i32, dbl, vp, vp, vp, vp, i64, C.POINTER(i64)]
    L.eacham_match_all_pairs_dev.argtypes = [vp, vp, i32, dbl, i32, i32, vp, vp, vp, i64, vp, vp]
    L.eacham_match_debug_batches.argtypes = [vp, i32, i32, vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.eacham_ba_solve.argtypes = [vp, vp, vp, vp]
    L.eacham_ba_prepare.argtypes = [vp, vp, C.POINTER(vp)]
    L.eacham_ba_run.argtypes = [vp, vp, vp, vp]
    L.eacham_ba_release.argtypes = [vp, vp]
    L.eacham_ba_release.restype = None
    L.eacham_ba_debug_step.argtypes = [vp, vp, dbl, vp, vp, vp, vp, vp, vp]
    L.eacham_ba_get_plan_info.argtypes = [vp, vp, vp]
    if hasattr(L, "eacham_ba_debug_structure"):
        L.eacham_ba_debug_structure.argtypes = [vp, vp, i32, vp, i64, C.POINTER(i64)]
    L.eacham_triangulate_tracks.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, C.c_float, C.c_float, vp, vp, vp]
    L.eacham_two_view_points.argtypes = [vp, i32, vp, vp, vp, i32, vp, C.c_float, C.c_float, i32, vp, vp, vp]
    L.eacham_score_hypotheses.argtypes = [vp, i32, i32, vp, vp, i32, vp, vp, C.c_float, vp, vp, vp]
    L.eacham_solve_minimal.argtypes = [vp, i32, i32, vp, vp, vp, i32, vp, vp, vp]
    L.eacham_solve_pnp.argtypes = [vp, i32, vp, vp, vp, i32, i32, vp, vp, vp]
    L.eacham_graph_best_pair.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    if hasattr(L, "eacham_graph_create"):
        L.eacham_graph_create.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, vp, C.POINTER(vp)]
        L.eacham_graph_destroy.argtypes = [vp]
        L.eacham_graph_destroy.restype = None
        L.eacham_graph_set_frame.argtypes = [vp, i32, i32, vp, i32]
        L.eacham_graph_set_frames.argtypes = [vp, i32, vp, vp, vp, vp]
        L.eacham_graph_query.argtypes = [vp, vp, i32, vp]
    L.eacham_reprojection_errors.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp]
    L.eacham_profile_enable.argtypes = [vp, i32]
    L.eacham_profile_reset.argtypes = [vp]
    L.eacham_profile_get.argtypes = [vp, i32, C.POINTER(i64), C.POINTER(dbl)]
    if hasattr(L, "eacham_order_pairs"):  # (absent from older diagnostic builds selected with EACHAM_HIP_LIB)
        L.eacham_order_pairs.argtypes = [vp, i32]
        L.eacham_shard_bounds.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    if hasattr(L, "eacham_comm_init"):
        L.eacham_comm_init.argtypes = [i32, vp, C.POINTER(vp)]
        L.eacham_comm_destroy.argtypes = [vp]
        L.eacham_comm_destroy.restype = None
        L.eacham_comm_last_error.argtypes = [vp]
        L.eacham_comm_last_error.restype = C.c_char_p
        L.eacham_comm_size.argtypes = [vp]
        L.eacham_comm_ctx.argtypes = [vp, i32]
        L.eacham_comm_ctx.restype = vp
        L.eacham_comm_upload_descriptors.argtypes = [vp, i32, vp, i32, i32]
        L.eacham_match_all_pairs_sharded.argtypes = [vp, vp, i32, dbl, i32, i32, vp, vp, vp, vp, i64, C.POINTER(i64)]
        L.eacham_assemble_match_graph.argtypes = [vp, vp, i32, i32, i32, i64, vp, vp, vp, vp, vp, i64, C.POINTER(i64)]
    if hasattr(L, "eacham_comm_match_run"):
        L.eacham_comm_match_run.argtypes = [vp, vp, i32, dbl, i32, i32, i32, C.POINTER(i64)]
        L.eacham_comm_match_fetch.argtypes = [vp, vp, vp, vp, vp, i64, C.POINTER(i64)]
        L.eacham_shard_bounds_weighted.argtypes = [i32, i32, vp, vp]
        L.eacham_assemble_match_graph_bounds.argtypes = [vp, vp, i32, i32, i32, i64, vp, vp, vp, vp, vp, vp, i64, C.POINTER(i64)]
        L.eacham_comm_edge_region.argtypes = [i32, vp, C.POINTER(i64)]
    _lib = L
    return L


# ---- bundle adjustment structs (include/eacham_hip.h) ---------------------------------------------
BA_LM, BA_DOGLEG = 0, 1
BA_DONE, BA_SKIPPED, BA_INDETERMINATE = 0, 1, 2
BA_LM_FACTOR_RESET, BA_LM_FACTOR_DOUBLE = 0, 1
BA_ORDER_AUTO, BA_ORDER_NATURAL, BA_ORDER_RCM, BA_ORDER_ND = 0, 1, 2, 3


class BaProblem(C.Structure):
    _fields_ = [("n_cams", C.c_int32), ("n_points", C.c_int32), ("n_obs", C.c_int32), ("ordering", C.c_int32),
                ("cam_T_wc", C.c_void_p), ("cam_fixed", C.c_void_p), ("points", C.c_void_p),
                ("point_observers", C.c_void_p), ("obs_cam", C.c_void_p), ("obs_point", C.c_void_p),
                ("obs_uv", C.c_void_p), ("K", C.c_double * 4)]


class BaPlanInfo(C.Structure):
    _fields_ = [("n_panels", C.c_int32), ("n_tiles", C.c_int32), ("n_levels", C.c_int32), ("ordering", C.c_int32),
                ("nd_leaf", C.c_int32), ("reserved", C.c_int32), ("tile_updates", C.c_int64), ("est_us", C.c_double), ("prepare_us", C.c_double * 3)]


class BaOptions(C.Structure):
    _fields_ = [("method", C.c_int32), ("max_iter", C.c_int32), ("max_tolerance", C.c_float),
                ("delta", C.c_float), ("use_preconditioner", C.c_int32), ("min_landmarks", C.c_int32),
                ("lm_factor_policy", C.c_int32), ("reserved", C.c_int32)]


class BaTraceRow(C.Structure):
    _fields_ = [("lambda_", C.c_double), ("new_error", C.c_double), ("lin_change", C.c_double),
                ("accepted", C.c_int32), ("outer", C.c_int32)]


class BaResult(C.Structure):
    _fields_ = [("cam_T_wc", C.c_void_p), ("points", C.c_void_p), ("K", C.c_double * 4),
                ("initial_error", C.c_double), ("final_error", C.c_double), ("final_lambda", C.c_double),
                ("status", C.c_int32), ("outer_iterations", C.c_int32), ("inner_iterations", C.c_int32),
                ("trace_cap", C.c_int32), ("trace_len", C.c_int32), ("reserved", C.c_int32),
                ("trace", C.POINTER(BaTraceRow))]
