"""CPU: the host-side IO of SURVEY.md §8(f) rank 4 — config schema, transform.json, transforms_nerf.json.
`eacham_amd/sfm_io.py` and `include/eacham/SfmIO.hpp` are held against each other byte for byte (the reference
writes with nlohmann::json, which is not in this tree: parity unpinned), and against the properties the
reference's code implies (key order, float widening, the kept quirks of SfmConfig::Parse)."""
import json
import math
import os
import subprocess

import numpy as np
import pytest

from eacham_amd import sfm_io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONFIG = {  # the schema of config/SfmConfigNerf.json
    "root_path": "/data/lego", "images_path": "/images", "transform_path": "/transform.json", "nerfy": True,
    "max_data_count": 0, "ui": True,
    "feature": {"min_features_count": 100, "max_features_count": 15000, "inliers_ratio": 0.8},
    "reconstruction": {
        "initial_pair": {"min_inliers": 350, "min_matches": 10, "min_corrs": 10, "max_reprojection_error": 3.5, "min_angle": 3.0},
        "processing": {"min_matches": 10, "min_corrs": 10, "max_reprojection_error": 8.0, "min_angle": 3.0, "min_pnp_inliers": 15}},
    "refine_ba": {"method": "LM", "max_iter": 100, "max_toler": 1e-5, "delta": 10.0, "use_preconditioner": False},
    "global_ba": {"method": "DogLeg", "max_iter": 50, "max_toler": 1e-4, "delta": 2.5, "use_preconditioner": True},
}


# The reference's own config files, copied as DATA (tests/golden/ref_*.json = /root/reference/config/SfmConfig.json
# and SfmConfigNerf.json, byte for byte): the only files the reference holds for any row of SURVEY.md §8.
# Expected values are what SfmConfig::Parse (modules/sfm/config/SfmConfig.h:27-71) makes of them, derived by hand.
REF_CONFIGS = {
    "ref_SfmConfig.json": dict(initial_min_inliers=450, initial_max_repr_error=4.0, initial_angle_deg=3.0, angle_deg=2.0,
                               global_iter=150, global_tol=1e-7),
    "ref_SfmConfigNerf.json": dict(initial_min_inliers=350, initial_max_repr_error=3.5, initial_angle_deg=3.0, angle_deg=3.0,
                                   global_iter=50, global_tol=1e-4),
}


def _f32(x):
    return float(np.float32(x))


def _check_reference_config(get, name):
    """`get(field)` reads a parsed field by its reference (C++) name."""
    e = REF_CONFIGS[name]
    assert get("imagesPath") == "path.../images" and get("outputTransformPath") == "path.../transform.json"  # root + path (:30-34)
    assert int(get("maxDataSize")) == 0 and int(get("ui")) == 0          # "ui": true is a boolean, never == "true" (:36)
    assert int(get("minFeaturesCount")) == 100 and int(get("maxFeaturesCount")) == 15000
    assert float(get("inliersRatio")) == _f32(0.8)
    assert int(get("initialMinInliers")) == e["initial_min_inliers"]
    assert float(get("initialMaxReprError")) == _f32(e["initial_max_repr_error"])
    # `float x = deg; x *= 3.141592 / 180.0;` : the product is formed in double and rounded to float (:47-48, :52-53)
    assert float(get("initialMinTriAngle")) == _f32(_f32(e["initial_angle_deg"]) * (3.141592 / 180.0))
    assert float(get("minTriAngle")) == _f32(_f32(e["angle_deg"]) * (3.141592 / 180.0))
    assert float(get("maxReprError")) == _f32(8.0) and int(get("minPnpInliers")) == 15
    assert (get("refine.method"), int(get("refine.maxIter"))) == ("LM", 100)
    assert float(get("refine.maxTolerance")) == _f32(1e-5) and float(get("refine.delta")) == 10.0
    assert int(get("refine.usePreconditioner")) == 0
    assert (get("global.method"), int(get("global.maxIter"))) == ("LM", e["global_iter"])
    assert float(get("global.maxTolerance")) == _f32(e["global_tol"])
    assert float(get("global.delta")) == 10.0 and int(get("global.usePreconditioner")) == 0   # read from refine_ba (:67-68)


@pytest.mark.parametrize("name", sorted(REF_CONFIGS))
def test_python_mirror_parses_the_reference_config_files(name):
    c = sfm_io.SfmConfig.parse(json.load(open(os.path.join(ROOT, "tests", "golden", name))))
    fields = {"imagesPath": c.images_path, "outputTransformPath": c.output_transform_path, "maxDataSize": c.max_data_size,
              "ui": c.ui, "minFeaturesCount": c.min_features_count, "maxFeaturesCount": c.max_features_count,
              "inliersRatio": c.inliers_ratio, "initialMinInliers": c.initial_min_inliers,
              "initialMaxReprError": c.initial_max_repr_error, "initialMinTriAngle": c.initial_min_tri_angle,
              "maxReprError": c.max_repr_error, "minTriAngle": c.min_tri_angle, "minPnpInliers": c.min_pnp_inliers}
    for pre, o in (("refine", c.refine_opt), ("global", c.global_opt)):
        fields.update({pre + ".method": o.method, pre + ".maxIter": o.max_iter, pre + ".maxTolerance": o.max_tolerance,
                       pre + ".delta": o.delta, pre + ".usePreconditioner": o.use_preconditioner})
    _check_reference_config(fields.__getitem__, name)
    # the OptimizerConfig the BA boundary receives (SfmConfig.h:15-22 -> eacham_ba_options)
    from eacham_amd import ba
    o = ba.c_options(ba.OptimizerConfig(c.global_opt.method, c.global_opt.max_iter, c.global_opt.max_tolerance,
                                        c.global_opt.delta, c.global_opt.use_preconditioner))
    assert (o.method, o.max_iter, o.use_preconditioner) == (0, REF_CONFIGS[name]["global_iter"], 0)
    assert o.max_tolerance == _f32(REF_CONFIGS[name]["global_tol"])


@pytest.mark.parametrize("name", sorted(REF_CONFIGS))
def test_cpp_header_parses_the_reference_config_files(tmp_path, name):
    import shutil
    exe = str(tmp_path / "io_driver")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "io_driver.cpp"), "-o", exe], check=True, capture_output=True)
    d = tmp_path / "case"
    d.mkdir()
    shutil.copy(os.path.join(ROOT, "tests", "golden", name), d / "config.json")
    r = subprocess.run([exe, str(d), "config-only"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [ln.split(" ", 1) for ln in open(d / "config.out").read().splitlines()]
    flat, seen = {}, 0
    for k, v in rows:   # the five OptimizerConfig names appear twice: refine first, then global
        if k in ("method", "maxIter", "maxTolerance", "delta", "usePreconditioner"):
            flat[("refine." if seen < 5 else "global.") + k] = v
            seen += 1
        else:
            flat[k] = v
    _check_reference_config(flat.__getitem__, name)


def _poses(n, seed=3):
    rng = np.random.default_rng(seed)
    out = {}
    for k in range(n):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        T = np.eye(4)
        T[:3, :3] = q * np.sign(np.linalg.det(q))
        T[:3, 3] = rng.normal(size=3) * 4
        out[7 * k + 2] = (f"./images/r_{k}.png", T)
    return out


def test_config_parse_keeps_the_reference_quirks():
    c = sfm_io.SfmConfig.parse(CONFIG)
    assert c.images_path == "/data/lego/images" and c.output_transform_path == "/data/lego/transform.json"
    assert c.ui is False                                  # a JSON boolean never equals the string "true"
    assert sfm_io.SfmConfig.parse({**CONFIG, "ui": "true"}).ui is True
    assert c.min_tri_angle == float(np.float32(3.0 * (3.141592 / 180.0)))  # `float *= double`: double product, float result
    assert abs(c.min_tri_angle - math.radians(3.0)) < 1e-7 and c.min_tri_angle != math.radians(3.0)
    assert c.inliers_ratio == float(np.float32(0.8)) and c.refine_opt.max_tolerance == float(np.float32(1e-5))
    assert (c.global_opt.method, c.global_opt.max_iter) == ("DogLeg", 50)
    assert c.global_opt.delta == 10.0 and c.global_opt.use_preconditioner is False     # taken from refine_ba
    with pytest.raises(KeyError):
        sfm_io.SfmConfig.parse({k: v for k, v in CONFIG.items() if k != "feature"})


def test_transform_json_layout(tmp_path):
    pos = _poses(3)
    path = str(tmp_path / "transform.json")
    sfm_io.save_positions(path, pos, 800, 800, 400.0, 400.0, 1111.1110311937682, 1111.1110311937682)
    text = open(path).read()
    doc = json.loads(text)
    assert text.endswith("}\n") and text.startswith('{\n    "camera_angle_x": ')
    assert list(doc) == sorted(doc)                       # nlohmann's std::map order
    assert doc["fl_x"] == float(np.float32(1111.1110311937682)) and doc["w"] == 800.0 and '"w": 800.0' in text
    assert doc["version"] == 0 and '"version": 0,' in text and doc["is_fisheye"] is False
    ax = float(np.float32(math.atan(800.0 / (float(np.float32(1111.1110311937682)) * 2.0)) * 2.0))
    assert doc["camera_angle_x"] == ax and doc["fovx"] == ax * 180.0 / 3.141592
    assert [f["file_path"] for f in doc["frames"]] == [pos[k][0] for k in sorted(pos)]
    for f, k in zip(doc["frames"], sorted(pos)):
        assert np.array_equal(np.array(f["transform_matrix"]), pos[k][1])   # shortest round-trip: exact
    assert json.loads(sfm_io.dumps(sfm_io.positions_document({}, 8, 8, 4, 4, 9, 9)))["frames"] is None


def test_transform_to_nerf(tmp_path):
    pos = _poses(4, seed=9)
    sfm_io.save_positions(str(tmp_path / "transform.json"), pos, 640, 480, 320, 240, 500, 500)
    out = sfm_io.transform_to_nerf(str(tmp_path))
    doc = json.load(open(out))
    assert os.path.basename(out) == "transforms_nerf.json" and doc["w"] == 640.0
    for f, k in zip(doc["frames"], sorted(pos)):
        M = np.array(f["transform_matrix"])
        want = np.linalg.inv(pos[k][1]) @ np.diag([1.0, -1.0, -1.0, 1.0])
        assert np.allclose(M, want, rtol=0, atol=1e-12)
        assert np.allclose(M[:3, 3], -pos[k][1][:3, :3].T @ pos[k][1][:3, 3], atol=1e-12)   # the camera centre


def test_cpp_header_writes_the_same_bytes(tmp_path):
    exe = str(tmp_path / "io_driver")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "io_driver.cpp"), "-o", exe], check=True, capture_output=True)
    d = tmp_path / "case"
    d.mkdir()
    json.dump(CONFIG, open(d / "config.json", "w"), indent=2)
    pos = _poses(5, seed=21)
    scal = [800, 600, 399.5, 301.25, 1111.1110311937682, 1107.3]
    with open(d / "positions.txt", "w") as f:
        f.write(f"{len(pos)}\n" + " ".join(float(v).hex() for v in scal) + "\n")
        for k in sorted(pos, reverse=True):  # any order in: std::map sorts
            f.write(f"{k} {pos[k][0]} " + " ".join(float(v).hex() for v in pos[k][1].ravel()) + "\n")
    rng = np.random.default_rng(5)
    nums = [0.0, -0.0, 1.0, -1.0, 0.1, 1e-4, 9.999e-5, 1e-5, 1.5e-7, 123456.789, 1e15, 1e16, 1.2345678901234567e17, 5e-324,
            1.7976931348623157e308, 800.0, float(np.float32(0.8)), 2.0 ** 53, 1e22, 1e21, 123456789012345678.0]
    nums += list(rng.normal(size=200) * 10.0 ** rng.integers(-12, 20, size=200))
    nums += [float(np.float32(v)) for v in rng.normal(size=50)]
    with open(d / "numbers.txt", "w") as f:
        f.write("\n".join(float(v).hex() for v in nums) + "\n")
    r = subprocess.run([exe, str(d)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # numbers
    got = open(d / "numbers.out").read().split("\n")[:-1]
    assert got == [sfm_io._num(float(v)) for v in nums]
    # documents
    py = tmp_path / "py"
    py.mkdir()
    sfm_io.save_positions(str(py / "transform.json"), pos, *scal)
    sfm_io.transform_to_nerf(str(py))
    assert open(d / "transform.json").read() == open(py / "transform.json").read()
    assert open(d / "transforms_nerf.json").read() == open(py / "transforms_nerf.json").read()
    # config
    c = sfm_io.SfmConfig.parse(CONFIG)
    want = [("imagesPath", c.images_path), ("outputTransformPath", c.output_transform_path),
            ("minFeaturesCount", c.min_features_count), ("maxFeaturesCount", c.max_features_count),
            ("inliersRatio", sfm_io._num(c.inliers_ratio)), ("maxDataSize", c.max_data_size),
            ("initialMinInliers", c.initial_min_inliers), ("initialMaxReprError", sfm_io._num(c.initial_max_repr_error)),
            ("initialMinTriAngle", sfm_io._num(c.initial_min_tri_angle)), ("maxReprError", sfm_io._num(c.max_repr_error)),
            ("minTriAngle", sfm_io._num(c.min_tri_angle)), ("minPnpInliers", c.min_pnp_inliers)]
    for o in (c.refine_opt, c.global_opt):
        want += [("method", o.method), ("maxIter", o.max_iter), ("maxTolerance", sfm_io._num(o.max_tolerance)),
                 ("delta", sfm_io._num(o.delta)), ("usePreconditioner", int(o.use_preconditioner))]
    want += [("ui", int(c.ui))]
    assert open(d / "config.out").read() == "".join(f"{k} {v}\n" for k, v in want)
