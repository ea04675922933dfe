"""CPU oracle loader (TEST INFRASTRUCTURE — never imported by the product package).

`lib()` builds the C restatement with gcc for the host it runs on (the .so is keyed by the CPU
model because it is compiled with -march=native) and returns the ctypes handle.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _cpu_tag() -> str:
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith(("model name", "flags")):
                    model += line
                    if line.startswith("flags"):
                        break
    except OSError:
        pass
    return hashlib.sha1(model.encode()).hexdigest()[:10]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, f"liboracle-{_cpu_tag()}.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        r = subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
        os.replace(os.path.join(_HERE, "liboracle.so"), so)
    return so


def build_sanitized() -> str:
    """ASAN + UBSAN build of the same sources (oracle/Makefile target liboracle-san.so)."""
    r = subprocess.run(["make", "-C", _HERE, "-B", "liboracle-san.so"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("sanitized oracle build failed:\n" + r.stdout + r.stderr)
    return os.path.join(_HERE, "liboracle-san.so")


def lib() -> ctypes.CDLL:
    """EACHAM_ORACLE_LIB selects another build of the same sources (the sanitizer run)."""
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(os.environ.get("EACHAM_ORACLE_LIB") or build())
    return _LIB
