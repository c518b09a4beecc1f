"""The spectrum widget's processing chain (friture/spectrum.py:125-184) without its Qt shell.

`SpectrumAnalyzer.handle_new_data(floatdata)` does what Spectrum_Widget's slot does per audio chunk:
push into the ring, transform every realizable frame (one batched launch of the float64 STFT kernel
over the contiguous ring window instead of a Python loop of analyzelive calls), exponential smoothing
across the new frames, dB + weighting (or the dual-channel ratio), spectral peak and the harmonic
product spectrum pitch — the last four fused in frt_spectrum_post.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .audioproc import audioproc
from .constants import SAMPLING_RATE
from .ringbuffer import RingBuffer
from .stft import StftEngine

DEFAULT_FFT_SIZE = 8192         # spectrum_settings.py:27-37
DEFAULT_RESPONSE_TIME = 0.025


class SpectrumAnalyzer:
    def __init__(self, fft_size: int = DEFAULT_FFT_SIZE, overlap: float = 3. / 4., weighting: int = 1,
                 response_time: float = DEFAULT_RESPONSE_TIME, dual_channels: bool = False):
        self._lib = _lib.init()
        self.ringbuffer = RingBuffer()
        self.proc = audioproc()
        self.overlap = overlap
        self.weighting = weighting
        self.dual_channels = dual_channels
        self.response_time = response_time
        self.old_index = 0
        self.setfftsize(fft_size)
        self.fmax = self.fpitch = 0.0
        self.dB_spectrogram = None

    # ---- configuration (names of friture/spectrum.py) ----------------------------------------------
    def setfftsize(self, fft_size):
        self.fft_size = fft_size
        self.proc.set_fftsize(fft_size)
        self.freq = self.proc.get_freq_scale()
        self.hop = int(fft_size * (1. - self.overlap))
        self._engine = StftEngine(fft_size, self.hop, 1, 64)
        self.update_weighting()
        self.dispbuffers1 = np.zeros(len(self.freq))
        self.dispbuffers2 = np.zeros(len(self.freq))
        self.setresponsetime(self.response_time)

    def setresponsetime(self, response_time):
        self.response_time = response_time
        w = 0.65
        n = response_time * SAMPLING_RATE / (self.fft_size * (1. - self.overlap))
        self.alpha = 1. - (1. - w) ** (1. / (n + 1))
        self.kernel = (1. - self.alpha) ** np.arange(2 * 4096 - 1, -1, -1)

    def setweighting(self, weighting):
        self.weighting = weighting
        self.update_weighting()

    def update_weighting(self):
        A, B, C = self.proc.get_freq_weighting()
        self.w = {0: np.zeros(A.shape), 1: A, 2: B}.get(self.weighting, C)

    # ---- the slot -------------------------------------------------------------------------------------
    def handle_new_data(self, floatdata):
        self.ringbuffer.push(floatdata, 0.)
        index = self.ringbuffer.offset
        available = index - self.old_index
        if available < 0:
            available = 0
            self.old_index = index
        needed = self.fft_size * (1. - self.overlap)
        realizable = int(np.floor(available / needed))
        if realizable <= 0:
            return None
        # frame i of the reference ends at old_index + i*hop: all frames form one contiguous window
        span = self.fft_size + (realizable - 1) * self.hop
        last = self.old_index + (realizable - 1) * self.hop
        window = self.ringbuffer.data_indexed(last, span)
        self.old_index += realizable * self.hop
        sp1, db, peak, pitch = self._post(self._engine.psd(window[0:1, :].copy())[0], self.dispbuffers1, self.w, None)
        self.dispbuffers1 = sp1
        if self.dual_channels and window.shape[0] > 1:
            sp2, db, peak, _ = self._post(self._engine.psd(window[1:2, :].copy())[0], self.dispbuffers2, None, sp1)
            self.dispbuffers2 = sp2
        self.dB_spectrogram = db
        self.fmax = self.freq[peak]
        self.fpitch = max(self.freq[pitch], 1e-20)
        return self.freq, db, self.fmax, self.fpitch

    def _post(self, psd, previous, weight, ref):
        psd = np.ascontiguousarray(psd, np.float64)             # [frames, bins]
        nf, nb = psd.shape
        prev = np.ascontiguousarray(previous, np.float64)
        sm, db = np.empty(nb), np.empty(nb)
        peak, pitch = ctypes.c_int(0), ctypes.c_int(0)
        wp = None if weight is None else np.ascontiguousarray(weight, np.float64)
        rp = None if ref is None else np.ascontiguousarray(ref, np.float64)
        _lib.check(self._lib.frt_spectrum_post(
            psd.ctypes.data, 0, nf, nb, nb, self.kernel.ctypes.data, len(self.kernel), float(self.alpha), prev.ctypes.data,
            None if wp is None else wp.ctypes.data, None if rp is None else rp.ctypes.data, sm.ctypes.data, db.ctypes.data,
            ctypes.byref(peak), ctypes.byref(pitch)))
        return sm, db, peak.value, pitch.value


class SpectrumAnalyzerStream(SpectrumAnalyzer):
    """The same slot with its spectra resident in HBM.  The samples' ring stays on the host (a push is a numpy copy: the
    widget completes a frame only every few chunks, and a device call per chunk would cost more than the ring is worth);
    when frames complete, their window goes up once through page-locked memory (frt_stft_run, host samples -> device
    spectra, no wait), and smoothing, dB + weighting, peak and harmonic-product pitch (frt_spectrum_post) read and write
    device buffers — the PSD frames and the smoothed spectra never leave the device; per completed frame one window goes
    up, the dB spectrum and two indices come down, one synchronisation."""

    def __init__(self, *args, **kw):
        import torch
        self._torch = torch
        self._dev = torch.device("cuda", torch.cuda.current_device())
        super().__init__(*args, **kw)

    def setfftsize(self, fft_size):
        super().setfftsize(fft_size)
        torch = self._torch
        nb = len(self.freq)
        self._d_disp1 = torch.zeros(nb, dtype=torch.float64, device=self._dev)
        self._d_disp2 = torch.zeros(nb, dtype=torch.float64, device=self._dev)
        self._d_next = torch.zeros(nb, dtype=torch.float64, device=self._dev)       # the smoothed spectrum being written
        self._d_psd = None                                                           # [frames, bins] of the last call
        self._db = np.empty(nb)

    def update_weighting(self):
        super().update_weighting()
        self._d_w = self._torch.from_numpy(np.ascontiguousarray(self.w, np.float64)).to(self._dev)

    def handle_new_data(self, floatdata):
        self.ringbuffer.push(floatdata, 0.)
        index = self.ringbuffer.offset
        available = index - self.old_index
        if available < 0:
            available = 0
            self.old_index = index
        needed = self.fft_size * (1. - self.overlap)
        realizable = int(np.floor(available / needed))
        if realizable <= 0:
            return None
        span = self.fft_size + (realizable - 1) * self.hop
        last = self.old_index + (realizable - 1) * self.hop
        window = self.ringbuffer.data_indexed(last, span)
        self.old_index += realizable * self.hop
        peak, pitch = self._post_dev(self._psd_dev(window[0], realizable), self._d_disp1, self._d_w, None)
        self._d_disp1, self._d_next = self._d_next, self._d_disp1
        if self.dual_channels and window.shape[0] > 1:
            peak, _ = self._post_dev(self._psd_dev(window[1], realizable), self._d_disp2, None, self._d_disp1)
            self._d_disp2, self._d_next = self._d_next, self._d_disp2
        db = self._db.copy()
        self.dB_spectrogram = db
        self.fmax = self.freq[peak]
        self.fpitch = max(self.freq[pitch], 1e-20)
        return self.freq, db, self.fmax, self.fpitch

    def _psd_dev(self, samples, n_frames):
        """PSD frames [n_frames, bins] of a host window, left on the device (enqueued on the engine's stream, not waited for).
        frt_spectrum_post launches on the null stream, which is ordered behind blocking streams only (friture_hip.h): the
        engine is put back on the null stream first, whatever a caller may have installed with frt_stft_set_stream."""
        torch = self._torch
        x = np.ascontiguousarray(samples, np.float64)
        nb = len(self.freq)
        if self._d_psd is None or self._d_psd.shape[0] < n_frames:
            self._d_psd = torch.empty((n_frames, nb), dtype=torch.float64, device=self._dev)
        nf = ctypes.c_int64(0)
        e = self._engine
        _lib.check(e._lib.frt_stft_set_stream(e._h, None))
        _lib.check(e._lib.frt_stft_run(e._h, 0, x.ctypes.data, x.shape[0], x.shape[0], ctypes.c_void_p(self._d_psd.data_ptr()),
                                       ctypes.byref(nf)))
        assert nf.value == n_frames
        return self._d_psd[:n_frames]

    def _post_dev(self, psd, disp, weight, ref):
        nf, nb = psd.shape
        peak, pitch = ctypes.c_int(0), ctypes.c_int(0)
        vp = ctypes.c_void_p
        _lib.check(self._lib.frt_spectrum_post(
            vp(psd.data_ptr()), 0, nf, nb, nb, self.kernel.ctypes.data, len(self.kernel), float(self.alpha), vp(disp.data_ptr()),
            None if weight is None else vp(weight.data_ptr()), None if ref is None else vp(ref.data_ptr()), vp(self._d_next.data_ptr()),
            self._db.ctypes.data, ctypes.byref(peak), ctypes.byref(pitch)))
        return peak.value, pitch.value
