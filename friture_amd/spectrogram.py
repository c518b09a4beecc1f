"""The spectrogram widget's processing chain (friture/spectrogram.py:131-177) without its Qt shell:
ring buffer, batched float64 STFT of every realizable frame, dB + weighting + normalisation, then the
Transform_Pipeline (frequency resampler -> online time resampler -> colour transform) to pixels."""
from __future__ import annotations

from fractions import Fraction

import numpy as np

from .audioproc import audioproc
from .constants import SAMPLING_RATE
from .plotting import frequency_scales as fscales
from .ringbuffer import RingBuffer
from .signal.color_tranform import Color_Transform
from .signal.frequency_resampler import Frequency_Resampler
from .signal.online_linear_2D_resampler import Online_Linear_2D_resampler
from .signal.transform_pipeline import Transform_Pipeline
from .stft import StftEngine

DEFAULT_FFT_SIZE = 4096        # spectrogram_settings.py:27-34
DEFAULT_TIMERANGE = 10.



def _torch_device():
    """(torch, the device the HIP library is bound to) when torch with a GPU is importable, else None: the device-resident hand-over
    between the transform and the fused pipeline call needs a device allocation, which this class borrows from torch; without
    torch the frames take the host path (one more round trip per chunk, same pixels)."""
    try:
        import torch
    except ImportError:
        return None
    if not torch.cuda.is_available():
        return None
    from . import _lib
    return torch, torch.device("cuda", _lib.bound_device)


class Spectrogram:
    def __init__(self, fft_size=DEFAULT_FFT_SIZE, overlap=Fraction(3, 4), spec_min=-140., spec_max=0., weighting=0,
                 scale=fscales.Mel, minfreq=20., maxfreq=20000., screen_width=800, screen_height=400,
                 timerange_s=DEFAULT_TIMERANGE):
        self.ringbuffer = RingBuffer()
        self.proc = audioproc()
        self.overlap_frac = Fraction(overlap)
        self.spec_min, self.spec_max, self.weighting = spec_min, spec_max, weighting
        self.screen_width, self.screen_height, self.timerange_s = screen_width, screen_height, timerange_s
        self.frequency_resampler = Frequency_Resampler(scale, minfreq, maxfreq, screen_height)
        self.screen_resampler = Online_Linear_2D_resampler()
        self.audio_pipeline = Transform_Pipeline([self.frequency_resampler, self.screen_resampler, Color_Transform()])
        self.old_index = 0
        self.setfftsize(fft_size)

    def setfftsize(self, fft_size):
        self.fft_size = fft_size
        self.proc.set_fftsize(fft_size)
        self.freq = self.proc.get_freq_scale()
        self.frequency_resampler.setfreq(self.freq)
        self.hop = int(fft_size * (1. - float(self.overlap_frac)))
        self._engine = StftEngine(fft_size, self.hop, 1, 64)
        A, B, C = self.proc.get_freq_weighting()
        self.w = {0: np.zeros(A.shape), 1: A, 2: B}.get(self.weighting, C)
        self._engine.set_epilogue(self.w, self.spec_min, self.spec_max, None)
        self.sfft_rate_frac = Fraction(SAMPLING_RATE, fft_size) / (Fraction(1) - self.overlap_frac) / 1000

    def handle_new_data(self, floatdata):
        self.ringbuffer.push(floatdata, 0.)
        index = self.ringbuffer.offset
        available = index - self.old_index
        if available < 0:
            available = 0
            self.old_index = index
        needed = self.fft_size * (1. - float(self.overlap_frac))
        realizable = int(np.floor(available / needed))
        if realizable <= 0:
            return None
        span = self.fft_size + (realizable - 1) * self.hop
        last = self.old_index + (realizable - 1) * self.hop
        window = self.ringbuffer.data_indexed(last, span)
        self.old_index += realizable * self.hop
        self.screen_resampler.set_height(self.screen_height)
        screen_rate_frac = Fraction(max(self.screen_width, 1), int(self.timerange_s * 1000))
        self.screen_resampler.set_ratio(self.sfft_rate_frac, screen_rate_frac)
        self.frequency_resampler.setnsamples(self.screen_height)
        if self.audio_pipeline.fusable() and _torch_device() is not None:
            # one wait per chunk: the frames' (dB + w - min)/(max - min) stay on the device, frame-major as the kernel writes them
            # (enqueued, not waited for), and the fused pipeline call reads them there (until round 4: two host round trips)
            return self.audio_pipeline.push_frames_device(self._norm_dev(window[0], realizable), len(self.freq), realizable)
        # (dB + w - min)/(max - min) per frame, reference layout (bins, frames)
        norm_spectrogram = self._engine.norm(window[0:1, :].copy())[0].T
        return self.audio_pipeline.push(norm_spectrogram)

    def _norm_dev(self, samples, n_frames):
        """Normalised dB frames [n_frames, bins] of a host window, left on the device.  frt_screen_columns launches on the null
        stream, which is ordered behind blocking streams only (friture_hip.h): the engine is put on the null stream first."""
        import ctypes

        from . import _lib
        from ._lib import FRT_STFT_NORM
        torch, device = _torch_device()
        x = np.ascontiguousarray(samples, np.float64)
        nb = len(self.freq)
        d = getattr(self, "_d_norm", None)
        if d is None or d.shape[0] < n_frames or d.shape[1] != nb:
            d = self._d_norm = torch.empty((max(n_frames, 8), nb), dtype=torch.float64, device=device)      # the ENGINE's device
        nf = ctypes.c_int64(0)
        e = self._engine
        _lib.check(e._lib.frt_stft_set_stream(e._h, None))
        _lib.check(e._lib.frt_stft_run(e._h, FRT_STFT_NORM, x.ctypes.data, x.shape[0], x.shape[0], ctypes.c_void_p(d.data_ptr()),
                                       ctypes.byref(nf)))
        assert nf.value == n_frames
        return ctypes.c_void_p(d.data_ptr())


class SpectrogramStream:
    """The same chain as ONE device-resident object (frt_specgram_*, specgram.hip): the samples' mirror ring, the spectra,
    the frequency map, the time resampler's carried column and the LUT live in HBM; a chunk costs one upload of its own
    samples, three launches and one download of the new pixel columns.  `handle_new_data` returns the block the widget
    hands to CanvasScaledSpectrogram.addData AFTER its flip of the frequency axis (spectrogram_image.py:82-92):
    uint32 [screen_height, columns], row 0 = highest frequency, or None when the chunk completed no frame."""

    def __init__(self, fft_size=DEFAULT_FFT_SIZE, overlap=Fraction(3, 4), spec_min=-140., spec_max=0., weighting=0,
                 scale=fscales.Mel, minfreq=20., maxfreq=20000., screen_width=800, screen_height=400,
                 timerange_s=DEFAULT_TIMERANGE, ring_length=None):
        import ctypes
        from . import _lib
        self._ct, self._libmod = ctypes, _lib
        self._lib = _lib.init()
        self.proc = audioproc()
        self.overlap_frac = Fraction(overlap)
        self.spec_min, self.spec_max, self.weighting = spec_min, spec_max, weighting
        self.scale, self.minfreq, self.maxfreq = scale, minfreq, maxfreq
        self.screen_width, self.screen_height, self.timerange_s = screen_width, screen_height, timerange_s
        self._h = ctypes.c_void_p()
        self._ring_length = ring_length
        self._colors = Color_Transform().colors
        self.setfftsize(fft_size)

    def _release(self):
        if self._h.value:
            self._lib.frt_specgram_destroy(self._h)
            self._h = self._ct.c_void_p()

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def setfftsize(self, fft_size):
        ct, check = self._ct, self._libmod.check
        self._release()
        self.fft_size = fft_size
        self.proc.set_fftsize(fft_size)
        self.freq = np.ascontiguousarray(self.proc.get_freq_scale(), np.float64)
        ring = self._ring_length or max(10000, 4 * fft_size)           # ringbuffer.py:34 starts at 10000 and grows on demand
        check(self._lib.frt_specgram_create(ct.byref(self._h), fft_size, float(self.overlap_frac), int(ring)))
        A, B, C = self.proc.get_freq_weighting()
        w = np.ascontiguousarray({0: np.zeros(A.shape), 1: A, 2: B}.get(self.weighting, C), np.float64)
        lut = np.ascontiguousarray(self._colors, np.uint32)
        DP, UP = ct.POINTER(ct.c_double), ct.POINTER(ct.c_uint32)
        check(self._lib.frt_specgram_set_epilogue(self._h, w.ctypes.data_as(DP), float(self.spec_min), float(self.spec_max),
                                                  lut.ctypes.data_as(UP)))
        self.sfft_rate_frac = Fraction(SAMPLING_RATE, fft_size) / (Fraction(1) - self.overlap_frac) / 1000
        self._screen = None

    def _sync_screen(self):
        ct, check = self._ct, self._libmod.check
        key = (self.screen_height, self.scale, self.minfreq, self.maxfreq)
        if key != self._screen:
            lo, hi = self.scale.transform(self.minfreq), self.scale.transform(self.maxfreq)
            targets = np.ascontiguousarray(self.scale.inverse(np.linspace(lo, hi, self.screen_height)), np.float64)
            DP = ct.POINTER(ct.c_double)
            check(self._lib.frt_specgram_set_screen(self._h, self.freq.ctypes.data_as(DP), targets.ctypes.data_as(DP), int(self.screen_height)))
            self._screen = key
        screen_rate_frac = Fraction(max(self.screen_width, 1), int(self.timerange_s * 1000))
        # set_ratio(L, M) divides float(L) / M (online_linear_2D_resampler.py:38): the same division on the other side
        check(self._lib.frt_specgram_set_ratio(self._h, float(self.sfft_rate_frac), float(screen_rate_frac)))

    def handle_new_data(self, floatdata):
        ct = self._ct
        x = np.ascontiguousarray(np.asarray(floatdata, np.float64)[0])
        self._sync_screen()
        frames_max = x.size // max(1, int(self.fft_size * (1. - float(self.overlap_frac)))) + 2
        ratio = float(self.sfft_rate_frac) / float(Fraction(max(self.screen_width, 1), int(self.timerange_s * 1000)))
        max_cols = int(frames_max / ratio) + 4
        out = np.empty((self.screen_height, max_cols), np.uint32)
        n_cols, n_frames = ct.c_int(0), ct.c_int(0)
        self._libmod.check(self._lib.frt_specgram_push(self._h, x.ctypes.data, x.size, out.ctypes.data, max_cols, ct.byref(n_cols),
                                                        ct.byref(n_frames)))
        if n_frames.value == 0:
            return None
        return out[:, :n_cols.value]
