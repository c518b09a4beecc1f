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


def exp_smoothed_value_groups(kernels, alphas, blocks, previous, square=False):
    """One launch for a list of exp_smoothed_value_2d calls (frt_exp_smooth_groups): group g = (kernels[g], alphas[g],
    blocks[g]); a block is a [rows_g, n_g] float64 array, or a tuple (first_row, rows_g) of a contiguous float64 row and the
    number of equally long rows that follow it back to back in memory (the packed band signals of Octave_Filters.filter).
    `previous`: one value per row, groups concatenated.  square: the data are squared on the device first (the octave-spectrum
    widget smooths y^2, friture/octavespectrum.py:103-112).  Every row equals its own exp_smoothed_value call bit for bit."""
    import ctypes
    G = len(blocks)
    ks = [np.ascontiguousarray(k, np.float64) for k in kernels]
    keep, ptrs, nfs, nts = [], [], [], []
    for b in blocks:
        if isinstance(b, tuple):
            row, nrows = b
            assert row.dtype == np.float64 and row.ndim == 1 and row.flags.c_contiguous
            keep.append(row)
            ptrs.append(row.ctypes.data)
            nfs.append(int(nrows))
            nts.append(row.shape[0])
        else:
            a = np.ascontiguousarray(b, np.float64)
            a = a.reshape(a.shape[0], -1) if a.ndim != 2 else a
            keep.append(a)
            ptrs.append(a.ctypes.data)
            nfs.append(a.shape[0])
            nts.append(a.shape[1])
    previous = np.ascontiguousarray(previous, np.float64)
    rows = sum(nfs)
    assert previous.shape == (rows,) and len(ks) == G and len(alphas) == G
    out = np.empty(rows, np.float64)
    kp = (ctypes.c_void_p * G)(*[k.ctypes.data for k in ks])
    dp = (ctypes.c_void_p * G)(*ptrs)
    nk = (ctypes.c_int * G)(*[k.shape[0] for k in ks])
    nf = (ctypes.c_int * G)(*nfs)
    nt = (ctypes.c_int * G)(*nts)
    rs = (ctypes.c_int64 * G)(*[max(n, 1) for n in nts])
    al = (ctypes.c_double * G)(*[float(a) for a in alphas])
    lib = _lib.init()
    _lib.check(lib.frt_exp_smooth_groups(G, kp, nk, al, dp, nf, nt, rs, int(bool(square)), previous.ctypes.data, out.ctypes.data))
    return out
