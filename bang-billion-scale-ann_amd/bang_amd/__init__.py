"""bang_amd -- Python (ctypes) binding of libbang.so, the MI355X-native BANG_Base search engine.

The product is the shared library (HIP kernels + C++ host engine, ``csrc/``); this package only
binds its C-ABI (``include/bang_c.h``) and carries the file-format / synthetic-index tooling.
There is no Python or CPU fallback: every compute call needs libbang.so and a HIP device.
"""
from .binding import (  # noqa: F401
    BangError, Engine, DeviceBuffer, IterState, lib, lib_path, build, device_count,
    U8, I8, F32, DIST_L2, DIST_MIPS, GRAPH_HOST, GRAPH_DEVICE, GRAPH_AUTO, DTYPE_CODE,
)
