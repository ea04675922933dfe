"""Host-side IO around the hot path: the reference's config schema and its
`transform.json` / `transforms_nerf.json` writers, so that a headless driver on top of this library reads
`config/SfmConfig*.json` and emits what eacham's tools emit. No GPU work here; C++ mirror: `include/eacham/SfmIO.hpp`.

  SfmConfig.parse      modules/sfm/config/SfmConfig.h:27-71 (quirks kept, see below)
  save_positions       modules/sfm/utils/Saver.h:13-73, called at apps/sfm/main.cpp:259-264
  transform_to_nerf    apps/sfm/TransformToNerf.cpp:40-66

The reference writes JSON with nlohmann::json (`file << std::setw(4) << j`): object keys in lexicographic
order, 4-space indent, arrays one element per line, floating-point numbers as the shortest string that
round-trips the double (integers-valued doubles with a trailing ".0"), `float` arguments widened to double
first. nlohmann itself is not in this tree (un-vendored Conan dependency): the format is restated from its
documented behaviour and **parity is unpinned**.
"""
from __future__ import annotations

import json
import math
from dataclasses import dataclass, field

import numpy as np


@dataclass
class OptimizerOptions:
    """OptimizerConfig of SfmConfig.h:15-22 (the BA options of include/eacham_hip.h carry the same fields)."""
    method: str = "LM"
    max_iter: int = 100
    max_tolerance: float = 1e-5
    delta: float = 10.0
    use_preconditioner: bool = False


def _f32(x) -> float:
    return float(np.float32(x))


@dataclass
class SfmConfig:
    images_path: str = ""
    output_transform_path: str = ""
    min_features_count: int = 0
    max_features_count: int = 0
    inliers_ratio: float = 0.0
    max_data_size: int = 0
    initial_min_inliers: int = 0
    initial_max_repr_error: float = 0.0
    initial_min_tri_angle: float = 0.0   # radians
    max_repr_error: float = 0.0
    min_tri_angle: float = 0.0           # radians
    min_pnp_inliers: int = 0
    refine_opt: OptimizerOptions = field(default_factory=OptimizerOptions)
    global_opt: OptimizerOptions = field(default_factory=OptimizerOptions)
    ui: bool = False

    @staticmethod
    def parse(data: dict) -> "SfmConfig":
        """SfmConfig::Parse, field by field. Kept on purpose (SURVEY Appendix C style quirks):
        * `ui` is true only for the STRING "true" (`data["ui"] == "true"`, :36): a JSON boolean gives false;
        * angles are converted with the literal 3.141592 in float arithmetic (:48-49, :53-54);
        * `global_ba` takes `delta` and `use_preconditioner` from `refine_ba` (:67-68);
        * float members hold float32 values."""
        c = SfmConfig()
        root = data["root_path"]
        c.images_path = root + data["images_path"]
        c.output_transform_path = root + data["transform_path"]
        c.max_data_size = int(data["max_data_count"])
        c.ui = data["ui"] == "true"
        feature = data["feature"]
        c.min_features_count = int(feature["min_features_count"])
        c.max_features_count = int(feature["max_features_count"])
        c.inliers_ratio = _f32(feature["inliers_ratio"])
        rec = data["reconstruction"]
        ini, proc = rec["initial_pair"], rec["processing"]
        c.initial_min_inliers = int(ini["min_inliers"])
        c.initial_max_repr_error = _f32(ini["max_reprojection_error"])
        # `float *= double`: the product is formed in double and rounded back to float
        c.initial_min_tri_angle = _f32(float(np.float32(ini["min_angle"])) * (3.141592 / 180.0))
        c.max_repr_error = _f32(proc["max_reprojection_error"])
        c.min_tri_angle = _f32(float(np.float32(proc["min_angle"])) * (3.141592 / 180.0))
        c.min_pnp_inliers = int(proc["min_pnp_inliers"])
        refine, glob = data["refine_ba"], data["global_ba"]
        c.refine_opt = OptimizerOptions(refine["method"], int(refine["max_iter"]), _f32(refine["max_toler"]),
                                        _f32(refine["delta"]), bool(refine["use_preconditioner"]))
        c.global_opt = OptimizerOptions(glob["method"], int(glob["max_iter"]), _f32(glob["max_toler"]),
                                        _f32(refine["delta"]), bool(refine["use_preconditioner"]))
        return c

    @staticmethod
    def load(path: str) -> "SfmConfig":
        with open(path) as f:
            return SfmConfig.parse(json.load(f))


# ---- nlohmann-style serialisation ------------------------------------------------------------------------
def _num(x) -> str:
    """A JSON number as nlohmann::json prints it: integers plainly, doubles as the shortest round-trip
    string with a '.0' when it would otherwise look like an integer; non-finite values become null."""
    if isinstance(x, bool):
        return "true" if x else "false"
    if isinstance(x, (int, np.integer)):
        return str(int(x))
    x = float(x)
    if not math.isfinite(x):
        return "null"
    return repr(x)  # CPython's repr IS shortest round-trip, with the same 'e-05' / '.0' conventions


def dumps(obj, indent: int = 4, _level: int = 0) -> str:
    """`std::setw(4) << json`: sorted keys, one array element per line, empty containers as {} / []."""
    pad, pad_in = " " * (indent * _level), " " * (indent * (_level + 1))
    if isinstance(obj, dict):
        if not obj:
            return "{}"
        items = [f"{pad_in}{json.dumps(str(k), ensure_ascii=False)}: {dumps(v, indent, _level + 1)}" for k, v in sorted(obj.items())]
        return "{\n" + ",\n".join(items) + "\n" + pad + "}"
    if isinstance(obj, (list, tuple)):
        if not obj:
            return "[]"
        return "[\n" + ",\n".join(pad_in + dumps(v, indent, _level + 1) for v in obj) + "\n" + pad + "]"
    if isinstance(obj, str):
        return json.dumps(obj, ensure_ascii=False)
    if obj is None:
        return "null"
    return _num(obj)


def positions_document(positions, w, h, cx, cy, fx, fy) -> dict:
    """The JSON object of SavePositions (Saver.h:20-63). `positions`: {frame id: (file path, 4x4 world->camera
    transform)} — written in ascending id (std::map), the matrix as given (the reference stores
    `Node::GetTransform()`), the six scalars as `float` arguments."""
    w, h, cx, cy, fx, fy = (_f32(v) for v in (w, h, cx, cy, fx, fy))
    ax = _f32(math.atan(w / (fx * 2.0)) * 2.0)  # const float angleX = std::atan(w / (fx * 2.0)) * 2.0
    ay = _f32(math.atan(h / (fy * 2.0)) * 2.0)
    doc = {"version": 0, "w": w, "h": h, "cx": cx, "cy": cy, "fl_x": fx, "fl_y": fy,
           "k1": 0, "k2": 0, "k3": 0, "k4": 0, "p1": 0, "p2": 0, "is_fisheye": False,
           "camera_angle_x": ax, "camera_angle_y": ay, "fovx": ax * 180.0 / 3.141592, "fovy": ay * 180.0 / 3.141592,
           "frames": None}  # `frames["frames"] = { }` is null until the first push_back
    for fid in sorted(positions):
        path, T = positions[fid]
        T = np.asarray(T, dtype=np.float64).reshape(4, 4)
        if doc["frames"] is None:
            doc["frames"] = []
        doc["frames"].append({"file_path": path, "transform_matrix": [[float(v) for v in row] for row in T]})
    return doc


def save_positions(path, positions, w, h, cx, cy, fx, fy) -> None:
    with open(path, "w") as f:
        f.write(dumps(positions_document(positions, w, h, cx, cy, fx, fy)) + "\n")


def pose_to_nerf(T) -> np.ndarray:
    """TransformToNerf.cpp:52-58: camera->world of the stored world->camera pose with the y and z camera axes
    flipped (OpenCV -> NeRF/OpenGL convention): inverse(T) * diag(1, -1, -1, 1)."""
    P = inverse4(T)
    P[:, 1] = -P[:, 1]
    P[:, 2] = -P[:, 2]
    return P


def inverse4(T) -> np.ndarray:
    """General 4x4 inverse by Gauss-Jordan with partial pivoting, operation for operation the routine of
    include/eacham/SfmIO.hpp (so that the two writers agree to the last bit; Eigen's own inverse differs in
    rounding, like any other)."""
    a = [[float(v) for v in row] + [1.0 if c == r else 0.0 for c in range(4)] for r, row in enumerate(np.asarray(T, dtype=np.float64).reshape(4, 4))]
    for col in range(4):
        piv = col
        for r in range(col + 1, 4):
            if abs(a[r][col]) > abs(a[piv][col]):
                piv = r
        if a[piv][col] == 0.0:
            raise ValueError("singular pose matrix")
        a[col], a[piv] = a[piv], a[col]
        d = a[col][col]
        a[col] = [v / d for v in a[col]]
        for r in range(4):
            if r != col:
                f = a[r][col]
                a[r] = [v - f * w for v, w in zip(a[r], a[col])]
    return np.array([row[4:] for row in a], dtype=np.float64)


def transform_to_nerf(folder: str) -> str:
    """Reads <folder>/transform.json, writes <folder>/transforms_nerf.json, returns its path."""
    if not folder.endswith("/"):
        folder += "/"
    with open(folder + "transform.json") as f:
        doc = json.load(f)
    for frame in doc["frames"]:
        frame["transform_matrix"] = [[float(v) for v in row] for row in pose_to_nerf(frame["transform_matrix"])]
    out = folder + "transforms_nerf.json"
    with open(out, "w") as f:
        f.write(dumps(doc) + "\n")
    return out
