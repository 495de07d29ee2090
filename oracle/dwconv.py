"""ctypes front-end of oracle/dwconv_oracle.c (TEST INFRASTRUCTURE ONLY).

numpy float32 in, numpy float64 out.  Reference definition: CUTLASS host reference
``Depsep_Fprop/_Dgrad/_Wgrad`` (cutlass/tools/util/include/cutlass/util/reference/host/convolution.h:160-235,
:327, :420) == ``F.conv2d(x, w, padding=k//2, groups=C)`` (test_correctness.py:8-9).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libslak_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "dwconv_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        fp, dp, i = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double), ctypes.c_int
        for name in ("slak_oracle_dwconv2d_fwd", "slak_oracle_dwconv2d_bwd_data"):
            fn = getattr(_lib, name); fn.restype = None; fn.argtypes = [fp, fp, dp, i, i, i, i, i, i]
        fn = _lib.slak_oracle_dwconv2d_bwd_filter; fn.restype = None; fn.argtypes = [fp, fp, dp, i, i, i, i, i, i]
        fn = _lib.slak_oracle_bf16_round_array; fn.restype = None; fn.argtypes = [fp, fp, ctypes.c_size_t]
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def _dims(x, w):
    N, C, H, W = x.shape
    assert w.shape[0] == C and w.shape[1] == 1, (x.shape, w.shape)
    return N, C, H, W, w.shape[2], w.shape[3]


def dwconv2d_fwd(x, w):
    """y = depthwise cross-correlation, stride 1, pad (kh//2, kw//2).  float64 result."""
    x, w = _f32(x), _f32(w)
    N, C, H, W, kh, kw = _dims(x, w)
    y = np.empty((N, C, H, W), np.float64)
    _load().slak_oracle_dwconv2d_fwd(_p(x, ctypes.c_float), _p(w, ctypes.c_float), _p(y, ctypes.c_double), N, C, H, W, kh, kw)
    return y


def dwconv2d_bwd_data(dy, w):
    dy, w = _f32(dy), _f32(w)
    N, C, H, W, kh, kw = _dims(dy, w)
    dx = np.empty((N, C, H, W), np.float64)
    _load().slak_oracle_dwconv2d_bwd_data(_p(dy, ctypes.c_float), _p(w, ctypes.c_float), _p(dx, ctypes.c_double), N, C, H, W, kh, kw)
    return dx


def dwconv2d_bwd_filter(dy, x, kh, kw):
    dy, x = _f32(dy), _f32(x)
    N, C, H, W = x.shape
    assert dy.shape == x.shape
    dw = np.empty((C, 1, kh, kw), np.float64)
    _load().slak_oracle_dwconv2d_bwd_filter(_p(dy, ctypes.c_float), _p(x, ctypes.c_float), _p(dw, ctypes.c_double), N, C, H, W, kh, kw)
    return dw


def bf16_round(a):
    """Round a float32 array to the nearest bfloat16 (ties to even), returned as float32."""
    a = _f32(a)
    out = np.empty_like(a)
    _load().slak_oracle_bf16_round_array(_p(a, ctypes.c_float), _p(out, ctypes.c_float), a.size)
    return out
