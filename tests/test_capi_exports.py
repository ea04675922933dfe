"""CPU: the C-ABI library loads and exports every symbol include/eacham_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            text = open(os.path.join(ROOT, "include", fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names |= set(re.findall(r"\b(eacham_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "eacham_match_all_pairs" in syms and "eacham_ctx_create" in syms and len(syms) >= 16


def test_library_exports_every_declared_symbol():
    from eacham_amd import capi
    L = capi.lib()  # raises ImportError if the extension was not built: no silent fallback
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    assert not missing, f"libeacham_hip.so lacks {missing}"
    assert b"gfx950" in L.eacham_version()


def test_no_device_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from eacham_amd import capi, HipContext
    with pytest.raises(capi.EachamError):
        HipContext(0)


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "eacham_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in src and "liboracle" not in src and "oracle/" not in src, fn
