"""CPU: the small containers of the adapters (include/eacham/FlatMap.hpp) — IdTable, the generation-stamped landmark id -> dense index
table that RefineBA's graph walk (include/eacham/ReferenceGlue.hpp, modules/sfm/reconstruction/BundleAdjuster.cpp:57-162) reuses from call to
call without clearing, and FlatMap — against std::unordered_map / std::map on random key streams: growth, reuse far below the grown
size, the stamp's wrap-around (tests/cpp/idtable_driver.cpp, compiled with g++ under -fsanitize=address,undefined)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_idtable_and_flatmap_against_the_standard_containers(tmp_path):
    exe = str(tmp_path / "idtable_driver")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Werror", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "idtable_driver.cpp")], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.returncode, r.stdout[-300:], r.stderr[-600:])
