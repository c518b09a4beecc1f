"""Drop-in for the reference's `friture.audioproc.audioproc` (friture/audioproc.py:27-96).

Presents the attribute surface the unchanged widgets use: `analyzelive`,
`norm_square`, `set_fftsize`, `set_maxfreq`, `get_freq_scale`, `get_freq_weighting` and the
attributes `.window .freq .fft_size .size_sq .A .B .C .maxfreq`.  The transform itself runs on
the GPU through `frt_stft_analyzelive_f64` (the float64 instance of kernel K1); constructing the
object fails when libfriture_hip.so or a gfx950 device is missing — there is no numpy path.
"""
from __future__ import annotations

import ctypes
import logging

import numpy as np

from . import _lib, tables
from .constants import SAMPLING_RATE

_DP = ctypes.POINTER(ctypes.c_double)


class audioproc:
    def __init__(self):
        self.logger = logging.getLogger(__name__)
        self._lib = _lib.init()
        self._plan = ctypes.c_void_p()
        # the reference starts from a placeholder configuration (audioproc.py:32-40)
        self.fft_size = 10
        self.maxfreq = 1.0
        self.size_sq = 1.0
        self.window = np.arange(0, 1)
        self.freq = np.linspace(0, SAMPLING_RATE / 2, 10)
        self.A = self.B = self.C = 0.0 * self.freq

    # ---- transform --------------------------------------------------------------------------
    def analyzelive(self, samples):
        """Power spectrum |rfft(samples * window)|^2 / fft_size^2 of one float64 frame."""
        frame = np.ascontiguousarray(samples, np.float64)
        if frame.shape != (self.fft_size,):
            raise ValueError(f"analyzelive expects {self.fft_size} samples, got shape {frame.shape}")
        if not self._plan.value:
            _lib.check(self._lib.frt_stft_create(ctypes.byref(self._plan), int(self.fft_size), int(self.fft_size), 1, 64))
        spectrum = np.empty(self.fft_size // 2 + 1, np.float64)
        _lib.check(self._lib.frt_stft_analyzelive_f64(self._plan, frame.ctypes.data_as(_DP),
                                                      spectrum.ctypes.data_as(_DP)))
        return spectrum

    def norm_square(self, fft):
        """|fft|^2 / fft_size^2 of a spectrum the caller computed itself (audioproc.py:49-50)."""
        return (fft * fft.conjugate()).real / self.size_sq

    # ---- configuration ----------------------------------------------------------------------
    def set_fftsize(self, fft_size):
        if fft_size == self.fft_size:
            return
        self.fft_size = fft_size
        self._release_plan()
        self._refresh_tables()

    def set_maxfreq(self, maxfreq):
        if maxfreq == self.maxfreq:
            return
        self.maxfreq = maxfreq
        self._refresh_tables()

    def get_freq_scale(self):
        return self.freq

    def get_freq_weighting(self):
        return self.A, self.B, self.C

    # ---- internals --------------------------------------------------------------------------
    def _refresh_tables(self):
        n = self.fft_size
        if len(self.freq) != n / 2 + 1:
            self.freq = tables.rfft_frequencies(n)
            self.A, self.B, self.C = tables.weighting_db(self.freq, floor=1e-50)
        self.window = tables.hann_symmetric(n)
        self.size_sq = float(n) ** 2

    def _release_plan(self):
        if self._plan.value:
            self._lib.frt_stft_destroy(self._plan)
            self._plan = ctypes.c_void_p()

    def __del__(self):
        try:
            self._release_plan()
        except Exception:
            pass
