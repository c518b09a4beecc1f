"""friture/signal/online_linear_2D_resampler.py:14-97 on the GPU: stateful linear resampling of
spectrogram columns from the STFT rate to the pixel rate.

The class keeps the reference's scalar bookkeeping (orig_index / resampled_index / ratio decide how
many pixel columns each pushed column emits and with which weights) and hands the arithmetic of a
whole push — out = data (1 - a) + old a for every emitted pixel column, linear_interp.py:57-60 —
to one launch of time_resample_kernel (frt_time_resample); a height change Fourier-resamples the carried column on the
device (frt_fourier_resample)."""
from __future__ import annotations

import math

import numpy as np

from .. import _lib
from .scipy_resample import resample


class Online_Linear_2D_resampler:
    def __init__(self, interp_factor_L=1, decim_factor_M=1, height=1):
        self._lib = _lib.init()
        self.interp_factor_L = interp_factor_L
        self.decim_factor_M = decim_factor_M
        self.resampling_ratio = float(interp_factor_L) / decim_factor_M
        self.height = height
        self.orig_index = 0.
        self.resampled_index = 0.
        self.old_data = np.zeros((self.height))

    def set_ratio(self, interp_factor_L, decim_factor_M):
        if self.interp_factor_L != interp_factor_L or self.decim_factor_M != decim_factor_M:
            self.interp_factor_L = interp_factor_L
            self.decim_factor_M = decim_factor_M
            self.resampling_ratio = float(interp_factor_L) / decim_factor_M
            self.orig_index = 0.
            self.resampled_index = 0.

    def set_height(self, height):
        if self.height != height:
            self.height = height
            self.orig_index = 0.
            self.resampled_index = 0.
            # the carried column is Fourier-resampled to the new height, as the reference does to avoid a black line
            # after a resize (online_linear_2D_resampler.py:45-55, scipy_resample.py:51-141)
            self.old_data = resample(self.old_data, self.height)

    def processable(self, m):
        return int(np.ceil((self.orig_index + m - (self.resampled_index + self.resampling_ratio)) / self.resampling_ratio))

    def advance(self, n_cols):
        """The scalar index bookkeeping of a push of n_cols columns, as in the reference (online_linear_2D_resampler.py:61-97):
        returns (columns the reference allocates, source column per emitted pixel column, its weight).  Same float64
        operations in the same order as the reference's numpy expressions (resampled_index + ratio * k, k = 1..n), on
        Python floats: a chunk advances by one or two columns, where array temporaries cost more than the arithmetic."""
        ceil = math.ceil
        ratio = self.resampling_ratio
        total = int(ceil((self.orig_index + n_cols - (self.resampled_index + ratio)) / ratio))
        src, weights = [], []
        orig, res = self.orig_index, self.resampled_index
        for j in range(n_cols):
            orig += 1.
            n = int(ceil((orig - (res + ratio)) / ratio))
            if n <= 0:
                continue
            last = res
            for k in range(1, n + 1):
                last = res + ratio * float(k)
                weights.append(orig - last)
                src.append(j)
            res = last
        self.orig_index, self.resampled_index = orig, res
        return total, np.array(src, np.int32), np.array(weights, np.float64)

    def push(self, data):
        data = np.ascontiguousarray(data, np.float64)
        self.set_height(data.shape[0])
        n_cols = data.shape[1]
        total, s, a = self.advance(n_cols)
        src = s
        out = np.zeros((self.height, max(total, 0)))
        if len(src):
            old = np.ascontiguousarray(self.old_data, np.float64)
            res = np.empty((self.height, len(src)))
            _lib.check(self._lib.frt_time_resample(data.ctypes.data, old.ctypes.data, self.height, n_cols, s.ctypes.data,
                                                   a.ctypes.data, len(src), res.ctypes.data))
            out[:, :len(src)] = res
        if n_cols:
            self.old_data = data[:, -1].copy()
        return out
