"""friture/signal/exp_smoothing.py:11-107 on the GPU: closed-form exponential smoothing of a block.

value = alpha * dot(kernel[Nk-N:], data[:N]) + previous * (1 - alpha)^N, one wavefront per row
(exp_smooth_kernel, frt_exp_smooth_2d).  numpy evaluates the dot product with BLAS, whose summation
order is unspecified: results agree to rounding (1e-13), not bit for bit."""
from __future__ import annotations

import numpy as np

from .. import _lib


def exp_smoothed_value_2d(kernel, alpha, data, previous):
    data = np.ascontiguousarray(data, np.float64)
    kernel = np.ascontiguousarray(kernel, np.float64)
    previous = np.ascontiguousarray(previous, np.float64)
    nf, nt = data.shape
    out = np.empty(nf, np.float64)
    lib = _lib.init()
    _lib.check(lib.frt_exp_smooth_2d(kernel.ctypes.data, kernel.shape[0], float(alpha), data.ctypes.data, nf, nt,
                                     max(nt, 1), previous.ctypes.data, out.ctypes.data))
    return out


def exp_smoothed_value(kernel, alpha, data, previous):
    data = np.ascontiguousarray(data, np.float64)
    if data.shape[0] == 0:
        return previous
    return float(exp_smoothed_value_2d(kernel, alpha, data[None, :], np.array([previous], np.float64))[0])
