"""pclean_amd — MI355X-native hot path for probcomp/PClean (per-row particle-Gibbs /
MH rejuvenation sweep + built-in noise-model densities) behind a C ABI.

Importing the package never touches the GPU; every compute entry point raises
`PCleanHipError` if libpclean_hip.so or a gfx950 device is missing (no CPU fallback).
"""
from ._lib import PCleanHipError, HipContext, load_library  # noqa: F401

__all__ = ["PCleanHipError", "HipContext", "load_library"]
