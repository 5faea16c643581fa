"""basic_dsp_amd -- MI355X (gfx950) backend for basic_dsp's time/frequency-domain vector ops.

The product is the C-ABI shared library basic_dsp_amd/lib/libbasic_dsp_hip.so
(include/basic_dsp_hip.h).  This package is the thin Python host mirror used by the tests, the
benchmark and the multi-GPU batch driver; it has no CPU fallback.
"""
from ._lib import BackendError, Graph, LIB_PATH, lib, last_error, require_gpu  # noqa: F401
from .vector import DspVec  # noqa: F401
from .matrix import DspMat  # noqa: F401

__all__ = ["DspVec", "DspMat", "Graph", "BackendError", "lib", "LIB_PATH", "last_error", "require_gpu"]
