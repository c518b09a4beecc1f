"""friture/signal/lfilter.py:85-147 on the GPU: `lfilter_float64_1D(b, a, x, zi) -> (y, zf)`.

Direct form II transposed, float64, explicit state in and out.  The device kernel replays the
reference's operation order without fused multiply-adds, so outputs and final states are
bit-identical to the reference's Python loop (and to scipy.signal.lfilter).
"""
from __future__ import annotations

import ctypes

import numpy as np

from .. import _lib

_DP = ctypes.POINTER(ctypes.c_double)


def _dp(a):
    return a.ctypes.data_as(_DP)


def lfilter_float64_1D(b, a, x, zi):
    b = np.ascontiguousarray(b, np.float64)
    a = np.ascontiguousarray(a, np.float64)
    x = np.ascontiguousarray(x, np.float64)
    zi = np.ascontiguousarray(zi, np.float64)
    assert b.shape[0] == a.shape[0], "a and b must be of the same shape"
    assert zi.shape[0] == b.shape[0] - 1
    lib = _lib.init()
    y = np.empty_like(x)
    zf = np.empty_like(zi)
    _lib.check(lib.frt_lfilter_f64(_dp(b), _dp(a), len(b), _dp(x), len(x), _dp(zi), _dp(y), _dp(zf)))
    return y, zf
