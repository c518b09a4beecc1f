"""friture/signal/correlation.py:24-43 on the GPU: `generalized_cross_correlation(d0, d1)`.

GCC-PHAT of two equally long float64 windows (kernel K5, frt_gcc_phat).  Like the reference, the
function removes the means *in place* on its arguments: the means come back from the device and are
subtracted from the caller's arrays, because the delay estimator's overlapping ring-buffer windows
rely on that side effect (friture/delay_estimator.py:125-132).
"""
from __future__ import annotations

import ctypes
import threading

import numpy as np

from .. import _lib

# One handle per window length, shared by every caller of the module-level function; a handle owns its staging and scratch
# buffers, so calls on it are serialised (ctypes drops the GIL for the duration of the C call: two threads asking for the same
# length would otherwise run on the same buffers — the reference has one GUI thread, tests/test_soak_gpu.py has four).
_plans: dict = {}
_plans_lock = threading.Lock()


def _plan(length: int, n_pairs: int = 1):
    key = (length, n_pairs)
    with _plans_lock:
        entry = _plans.get(key)
        if entry is None:
            lib = _lib.init()
            h = ctypes.c_void_p()
            _lib.check(lib.frt_gcc_create(ctypes.byref(h), length, n_pairs))
            entry = _plans[key] = (h, threading.Lock())
    return entry


def generalized_cross_correlation(d0, d1):
    if len(d0) != len(d1):
        raise ValueError("operands could not be broadcast together with shapes (%d,) (%d,)" % (len(d0), len(d1)))
    lib = _lib.init()
    a0 = np.ascontiguousarray(d0, np.float64)
    a1 = np.ascontiguousarray(d1, np.float64)
    h, busy = _plan(len(a0))
    xcorr = np.empty(len(a0), np.float64)
    means = np.empty(2, np.float64)
    with busy:
        _lib.check(lib.frt_gcc_phat(h, a0.ctypes.data, a1.ctypes.data, xcorr.ctypes.data, None, means.ctypes.data))
    # the reference's in-place side effect on the caller's buffers
    d0 -= means[0]
    d1 -= means[1]
    return xcorr


class GccPhat:
    """Batched GCC-PHAT + read-out over `n_pairs` windows of `length` samples."""

    def __init__(self, length: int, n_pairs: int = 1):
        self._lib = _lib.init()
        self.length, self.n_pairs = length, n_pairs
        self._h = ctypes.c_void_p()
        _lib.check(self._lib.frt_gcc_create(ctypes.byref(self._h), length, n_pairs))

    def __del__(self):
        try:
            if self._h.value:
                self._lib.frt_gcc_destroy(self._h)
        except Exception:
            pass

    def correlate(self, d0, d1):
        """d0, d1: [n_pairs, length] float64 (numpy or torch CUDA).  Returns (xcorr, argmax)."""
        if type(d0).__module__.startswith("torch"):
            import torch
            assert d0.is_cuda and d0.dtype == torch.float64 and d0.is_contiguous() and d1.is_contiguous()
            out = torch.empty_like(d0)
            am = torch.empty(self.n_pairs, dtype=torch.int32, device=d0.device)
            self.means = torch.empty((self.n_pairs, 2), dtype=torch.float64, device=d0.device)      # of d0 / d1, per pair
            _lib.check(self._lib.frt_gcc_set_stream(self._h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
            _lib.check(self._lib.frt_gcc_phat(self._h, ctypes.c_void_p(d0.data_ptr()), ctypes.c_void_p(d1.data_ptr()),
                                              ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(am.data_ptr()),
                                              ctypes.c_void_p(self.means.data_ptr())))
            return out, am
        d0 = np.ascontiguousarray(d0, np.float64).reshape(self.n_pairs, self.length)
        d1 = np.ascontiguousarray(d1, np.float64).reshape(self.n_pairs, self.length)
        out = np.empty_like(d0)
        am = np.empty(self.n_pairs, np.int32)
        _lib.check(self._lib.frt_gcc_phat(self._h, d0.ctypes.data, d1.ctypes.data, out.ctypes.data, am.ctypes.data, None))
        return out, am

    def correlate_windows(self, p0, p1, device, stream):
        """The same for windows given as device POINTERS (ctypes.c_void_p; frt_delay_window), on HIP stream `stream`
        (ctypes.c_void_p).  Returns the correlation as a torch tensor [n_pairs, length]; the means stay in self.means."""
        import torch
        out = torch.empty((self.n_pairs, self.length), dtype=torch.float64, device=device)
        self.means = torch.empty((self.n_pairs, 2), dtype=torch.float64, device=device)
        _lib.check(self._lib.frt_gcc_set_stream(self._h, stream))
        _lib.check(self._lib.frt_gcc_phat(self._h, p0, p1, ctypes.c_void_p(out.data_ptr()), None, ctypes.c_void_p(self.means.data_ptr())))
        return out

    def readout(self, xcorr, old_smoothed, sample_rate, delayrange_s, alpha=0.3, stream=None):
        """Smoothing + peak pick + delay / confidence (delay_estimator.py:134-176).
        Returns (smoothed [n_pairs, length], list of DelayReadout)."""
        if type(xcorr).__module__.startswith("torch"):
            import torch
            assert xcorr.is_cuda and xcorr.dtype == torch.float64 and xcorr.is_contiguous()
            sm = torch.empty_like(xcorr)
            ro = (_lib.DelayReadout * self.n_pairs)()
            _lib.check(self._lib.frt_gcc_set_stream(self._h, stream if stream is not None
                                                    else ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
            _lib.check(self._lib.frt_gcc_readout(self._h, ctypes.c_void_p(xcorr.data_ptr()),
                                                 None if old_smoothed is None else ctypes.c_void_p(old_smoothed.data_ptr()), alpha,
                                                 float(sample_rate), float(delayrange_s), ctypes.c_void_p(sm.data_ptr()), ctypes.byref(ro)))
            return sm, list(ro)
        x = np.ascontiguousarray(xcorr, np.float64).reshape(self.n_pairs, self.length)
        old = None if old_smoothed is None else np.ascontiguousarray(old_smoothed, np.float64).reshape(x.shape)
        sm = np.empty_like(x)
        ro = (_lib.DelayReadout * self.n_pairs)()
        _lib.check(self._lib.frt_gcc_readout(self._h, x.ctypes.data, None if old is None else old.ctypes.data, alpha,
                                             float(sample_rate), float(delayrange_s), sm.ctypes.data, ctypes.byref(ro)))
        return sm, list(ro)
