"""The octave-spectrum widget's processing chain (friture/octavespectrum.py:91-156) without Qt:
Octave_Filters.filter (FFT overlap-add bank on the GPU), per-band exponential smoothing of y^2
(frt_exp_smooth_2d, one launch per decimation class) and 10 log10(sp + 1e-30) + weighting."""
from __future__ import annotations

import numpy as np

from .constants import NOCTAVE, SAMPLING_RATE
from .octavefilters import Octave_Filters
from .signal.exp_smoothing import exp_smoothed_value_2d

DEFAULT_BANDSPEROCTAVE = 3      # octavespectrum_settings.py:25-31
DEFAULT_RESPONSE_TIME = 1.


class OctaveSpectrum:
    def __init__(self, bandsperoctave: int = DEFAULT_BANDSPEROCTAVE, weighting: int = 1,
                 response_time: float = DEFAULT_RESPONSE_TIME):
        self.filters = Octave_Filters(bandsperoctave)
        self.weighting = weighting
        self.dispbuffers = [0] * bandsperoctave * NOCTAVE
        self.setresponsetime(response_time)

    def setresponsetime(self, response_time):
        self.response_time = response_time
        w = 0.65
        decs = self.filters.get_decs()
        ns = [response_time * SAMPLING_RATE / dec for dec in decs]
        Ns = [2 * 4096 / dec for dec in decs]
        self.alphas = [1. - (1. - w) ** (1. / (n + 1)) for n in ns]
        self.kernels = [(1. - alpha) ** np.arange(N - 1, -1, -1) for alpha, N in zip(self.alphas, Ns)]

    def setbandsperoctave(self, bandsperoctave):
        self.filters.setbandsperoctave(bandsperoctave)
        self.dispbuffers = [0] * bandsperoctave * NOCTAVE
        self.setresponsetime(self.response_time)

    def handle_new_data(self, floatdata):
        if floatdata.shape[1] == 0:
            return None
        y, _ = self.filters.filter(floatdata[0, :])
        bpo = self.filters.bandsperoctave
        sp = np.empty(len(y))
        for octave in range(NOCTAVE):            # bands of one octave share kernel, alpha and length
            lo = octave * bpo
            block = np.stack([band ** 2 for band in y[lo:lo + bpo]])
            sp[lo:lo + bpo] = exp_smoothed_value_2d(self.kernels[lo], self.alphas[lo], block,
                                                    np.asarray(self.dispbuffers[lo:lo + bpo], np.float64))
        self.dispbuffers = list(sp)
        w = {0: 0., 1: self.filters.A, 2: self.filters.B}.get(self.weighting, self.filters.C)
        db_spectrogram = 10 * np.log10(sp + 1e-30) + w
        return self.filters.flow, self.filters.fhigh, self.filters.f_nominal, db_spectrogram
