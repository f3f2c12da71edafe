"""CPU-side checks of the drop-in boundary: libmirx.so loads, exports every symbol that
include/mirx.h declares, and refuses to work without a GPU (no silent fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "mirx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mirx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_bound_and_exported():
    import mirx._lib as L
    names = _header_functions()
    assert len(names) >= 15
    assert sorted(L.SYMBOLS) == names, "ctypes table and include/mirx.h disagree"
    lib = L.load()                      # raises if any symbol is missing from the .so
    assert lib.mirx_version() == 100
    for n in names:
        assert hasattr(lib, n)


def test_bad_arguments_report_errors_without_gpu():
    import ctypes
    import mirx._lib as L
    lib = L.load()
    h = ctypes.c_void_p()
    rc = lib.mirx_index_create(0, 0, 0, ctypes.byref(h))      # dim 0 is rejected before any HIP call
    assert rc == -1 and b"dim" in lib.mirx_last_error()
    rc = lib.mirx_index_create(16, 7, 0, ctypes.byref(h))
    assert rc == -1 and b"metric" in lib.mirx_last_error()
    assert lib.mirx_index_size(None) == -1


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mirx.index import FlatIndex
    from mirx._lib import MirxError
    with pytest.raises(MirxError):
        FlatIndex(64)


def test_product_never_imports_oracle():
    """The package must not reference oracle/ anywhere (tier rule: oracle is test-only)."""
    pkg = os.path.join(ROOT, "image-retrieval---thesis-2026_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "oracle/_build" not in src and "liboracle" not in src, f
