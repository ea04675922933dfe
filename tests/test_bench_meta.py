"""bench.py's stamping of measured figures: a traffic figure is reported only with the kernel
sources it was measured on. CPU-only: no GPU, no compute."""
import json
import os
import shutil

import bench


def _copy_sources(tmp_path):
    src = os.path.join(bench.ROOT, "eacham_amd", "csrc")
    dst = tmp_path / "eacham_amd" / "csrc"
    dst.mkdir(parents=True)
    for fn in os.listdir(src):
        if fn.endswith((".hip", ".hpp")):
            shutil.copy(os.path.join(src, fn), dst / fn)
    return dst


def test_kernel_sha_follows_code_not_comments(tmp_path, monkeypatch):
    dst = _copy_sources(tmp_path)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    base = bench.kernel_source_sha()
    assert base == bench.kernel_source_sha()
    with open(dst / "score.hip", "a") as f:
        f.write("\n// a remark\n/* and\n   another */\n\n")
    assert bench.kernel_source_sha() == base          # comments and white space: the same kernels
    with open(dst / "score.hip", "a") as f:
        f.write("static int one_more_token;\n")
    assert bench.kernel_source_sha() != base          # a token: other kernels


def test_stale_traffic_profile_is_not_reported(tmp_path, monkeypatch):
    _copy_sources(tmp_path)
    (tmp_path / "profiles").mkdir()
    entry = {"FETCH_SIZE_KB_mean_per_dispatch": 1000.0, "WRITE_SIZE_KB_mean_per_dispatch": 500.0}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    good = {"__meta__": {"kernel_source_sha": bench.kernel_source_sha(), "command": "test"}, "eacham::some_kernel grid=256": entry}
    name = f"{bench.PROFILE_ROUND}_pmc_hbm_traffic.json"
    with open(tmp_path / "profiles" / name, "w") as f:
        json.dump(good, f)
    traffic, src = bench.measured_traffic("eacham::some_kernel")
    assert traffic == (2 * 1000.0 + 500.0) * 1024.0 and name in src
    good["__meta__"]["kernel_source_sha"] = "0123456789abcdef"
    with open(tmp_path / "profiles" / name, "w") as f:
        json.dump(good, f)
    traffic, src = bench.measured_traffic("eacham::some_kernel")
    assert traffic is None and src.startswith("stale")


def test_a_change_to_the_bundle_adjuster_does_not_disown_the_matcher_counters(tmp_path, monkeypatch):
    """Profiles record one sha per source group: the matcher's traffic figure survives an edit of ba.hip, not one of
    matcher.hip."""
    dst = _copy_sources(tmp_path)
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    entry = {"FETCH_SIZE_KB_mean_per_dispatch": 10.0, "WRITE_SIZE_KB_mean_per_dispatch": 5.0}
    meta = {"kernel_source_sha": bench.kernel_source_sha(), "kernel_source_sha_match": bench.kernel_source_sha("match"),
            "kernel_source_sha_ba": bench.kernel_source_sha("ba")}
    with open(tmp_path / "profiles" / f"{bench.PROFILE_ROUND}_pmc_hbm_traffic.json", "w") as f:
        json.dump({"__meta__": meta, "eacham::match_tile_kernel<8, 2> grid=256": entry}, f)
    with open(dst / "ba.hip", "a") as f:
        f.write("static int ba_token;\n")
    assert bench.measured_traffic("eacham::match_tile_kernel<8, 2>")[0] is not None
    with open(dst / "matcher.hip", "a") as f:
        f.write("static int matcher_token;\n")
    assert bench.measured_traffic("eacham::match_tile_kernel<8, 2>")[0] is None
