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
        # (dB + w - min)/(max - min) per frame, reference layout (bins, frames)
        norm_spectrogram = self._engine.norm(window[0:1, :].copy())[0].T
        self.screen_resampler.set_height(self.screen_height)
        screen_rate_frac = Fraction(max(self.screen_width, 1), int(self.timerange_s * 1000))
        self.screen_resampler.set_ratio(self.sfft_rate_frac, screen_rate_frac)
        self.frequency_resampler.setnsamples(self.screen_height)
        return self.audio_pipeline.push(norm_spectrogram)
