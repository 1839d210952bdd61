"""ctypes loader of tools/probe/libmaxsim_probe.so (include/maxsim_probe.h): measurement aids, never the product.

bench.py and the tools/ scripts use it to measure what THIS machine delivers for the kernels' access pattern and MFMA mix.
`lib()` returns None when the library has not been built (`make -C tools/probe`; __graft_entry__.build() does it)."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmaxsim_probe.so")
_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        return None
    import torch  # noqa: F401  -- maps the HIP runtime the library binds to

    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    L.msim_probe_last_error.restype = ctypes.c_char_p
    L.msim_probe_stream.argtypes = [i32, vp, i64, i32, vp, vp]
    L.msim_probe_stream.restype = i32
    L.msim_probe_mfma.argtypes = [i32, vp, i64, i32, vp, vp]
    L.msim_probe_mfma.restype = i32
    _lib = L
    return L
