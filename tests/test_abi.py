"""CPU tests of the boundary: the shared object loads and exports every symbol
declared in include/pclean_hip.h; compute without a GPU fails loudly."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "pclean_hip.h")).read()
    return sorted(set(re.findall(r"\b(pclean_[a-z0-9_]+)\s*\(", src)) - {"pclean_ctx"})


def test_library_exports_every_declared_symbol():
    from pclean_amd import build
    build.build(verbose=False)
    lib = ctypes.CDLL(build.LIB)
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/pclean_hip.h but not exported"


def test_no_cpu_fallback():
    import torch
    from pclean_amd import HipContext, PCleanHipError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(PCleanHipError):
        HipContext(0)


def test_struct_layouts_match_header():
    from pclean_amd import _lib
    assert ctypes.sizeof(_lib.Term) == 32 == _lib.TERM_DTYPE.itemsize
    assert ctypes.sizeof(_lib.Node) == 48 == _lib.NODE_DTYPE.itemsize
    assert ctypes.sizeof(_lib.InferConfig) == 28
