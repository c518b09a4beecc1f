"""friture/signal/frequency_resampler.py:26-83 on the GPU: resample spectrogram columns from the
FFT bins onto the screen rows of a frequency scale (kernel freq_resample_kernel, frt_freq_resample)."""
from __future__ import annotations

import numpy as np

from .. import _lib
from ..plotting import frequency_scales as fscales


class Frequency_Resampler:
    def __init__(self, scale=fscales.Linear, minfreq: float = 20., maxfreq: float = 20000., nsamples: int = 1) -> None:
        self._lib = _lib.init()
        self.scale = scale
        self.minfreq, self.maxfreq, self.nsamples = minfreq, maxfreq, nsamples
        self.freq = np.zeros((1))
        self.update_xscale()

    def setfreqrange(self, minfreq: float, maxfreq: float) -> None:
        self.minfreq, self.maxfreq = minfreq, maxfreq
        self.update_xscale()

    def update_xscale(self) -> None:
        lo, hi = self.scale.transform(self.minfreq), self.scale.transform(self.maxfreq)
        self.xscaled = self.scale.inverse(np.linspace(lo, hi, self.nsamples))

    def setnsamples(self, nsamples):
        if self.nsamples != nsamples:
            self.nsamples = nsamples
            self.update_xscale()

    def setfreqscale(self, scale) -> None:
        if scale != self.scale:
            self.scale = scale
            self.update_xscale()

    def setfreq(self, freq) -> None:
        self.freq = freq
        self.update_xscale()

    def push(self, data):
        """data: (bins, columns) float64 -> (nsamples, columns)."""
        data = np.ascontiguousarray(data, np.float64)
        freq = np.ascontiguousarray(self.freq, np.float64)
        targets = np.ascontiguousarray(self.xscaled, np.float64)
        if data.shape[0] != freq.size:
            raise ValueError("fp and xp are not of the same length.")          # numpy.interp's complaint
        out = np.zeros((targets.size, data.shape[1]))
        _lib.check(self._lib.frt_freq_resample(freq.ctypes.data, freq.size, targets.ctypes.data, targets.size,
                                               data.ctypes.data, data.shape[1], out.ctypes.data))
        return out
