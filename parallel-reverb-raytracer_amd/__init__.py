"""MI355X-native acoustic ray tracer: Python host side (ctypes over the C-ABI of include/rvb_capi.h).

The compute lives in csrc/ (hand-written HIP for gfx950) behind librvb_hip.so; this package only
marshals numpy / torch buffers into that C-ABI for tests and bench.py.  There is no CPU fallback:
importing `capi` raises if the shared library has not been built.
"""
from . import dtypes, scenes  # noqa: F401
