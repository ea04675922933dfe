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
    with open(tmp_path / "profiles" / "r02_pmc_hbm_traffic.json", "w") as f:
        json.dump(good, f)
    traffic, src = bench.measured_traffic("eacham::some_kernel")
    assert traffic == (2 * 1000.0 + 500.0) * 1024.0 and "r02_pmc_hbm_traffic.json" in src
    good["__meta__"]["kernel_source_sha"] = "0123456789abcdef"
    with open(tmp_path / "profiles" / "r02_pmc_hbm_traffic.json", "w") as f:
        json.dump(good, f)
    traffic, src = bench.measured_traffic("eacham::some_kernel")
    assert traffic is None and src.startswith("stale")
