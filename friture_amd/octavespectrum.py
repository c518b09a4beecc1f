"""The octave-spectrum widget's processing chain (friture/octavespectrum.py:91-156) without Qt:
Octave_Filters.filter (FFT overlap-add bank on the GPU), per-band exponential smoothing of y^2
(frt_exp_smooth_groups: all bands of a chunk in one launch) and 10 log10(sp + 1e-30) + weighting."""
from __future__ import annotations

import numpy as np

from .constants import NOCTAVE, SAMPLING_RATE
from .octavefilters import Octave_Filters
from .signal.exp_smoothing import exp_smoothed_value_groups

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
        # The reference smooths band by band (octavespectrum.py:103-112); bands of one octave share kernel, alpha and length and lie
        # back to back in the bank's packed output, so the chunk's 9 x bpo values are ONE call (frt_exp_smooth_groups, y^2 formed on
        # the device; until round 4: one call per octave, nine device round trips per chunk).
        blocks = []
        if getattr(self.filters, "_packed", None) is not None and y[0].base is self.filters._packed:
            blocks = [(y[octave * bpo], bpo) for octave in range(NOCTAVE)]      # this call's packed output: octaves are [bpo, m] blocks
        else:
            for octave in range(NOCTAVE):
                lo = octave * bpo
                row, m = y[lo], y[lo].shape[0]
                packed = all(y[lo + i].ctypes.data == row.ctypes.data + i * m * 8 and y[lo + i].shape[0] == m for i in range(1, bpo))
                blocks.append((row, bpo) if packed and row.flags.c_contiguous else np.stack(y[lo:lo + bpo]))
        sp = exp_smoothed_value_groups([self.kernels[o * bpo] for o in range(NOCTAVE)], [self.alphas[o * bpo] for o in range(NOCTAVE)],
                                       blocks, np.asarray(self.dispbuffers, np.float64), square=True)
        self.dispbuffers = list(sp)
        w = {0: 0., 1: self.filters.A, 2: self.filters.B}.get(self.weighting, self.filters.C)
        db_spectrogram = 10 * np.log10(sp + 1e-30) + w
        return self.filters.flow, self.filters.fhigh, self.filters.f_nominal, db_spectrogram


class OctaveSpectrumStream(OctaveSpectrum):
    """The same chain with everything between the chunk and the 9 x bpo dB values on the device: the FIR bank's tails, the
    smoothed energies and the weighting live in the bank object (frt_octbank_energies in mode 1, one block = the chunk);
    a chunk costs one upload of its samples and one download of the band vector.  Values are float32 on the way out
    (1e-7 relative; the north star's band-energy tolerance is 1e-5)."""

    def __init__(self, bandsperoctave: int = DEFAULT_BANDSPEROCTAVE, weighting: int = 1,
                 response_time: float = DEFAULT_RESPONSE_TIME):
        super().__init__(bandsperoctave, weighting, response_time)
        from .filter import FirBank
        self._bank = FirBank(bandsperoctave, 1)

    def setbandsperoctave(self, bandsperoctave):
        super().setbandsperoctave(bandsperoctave)
        from .filter import FirBank
        self._bank = FirBank(bandsperoctave, 1)

    def handle_new_data(self, floatdata):
        n = floatdata.shape[1]
        if n == 0:
            return None
        w = {0: np.zeros(len(self.alphas)), 1: self.filters.A, 2: self.filters.B}.get(self.weighting, self.filters.C)
        x = np.ascontiguousarray(floatdata[0:1, :], np.float32)
        db = self._bank.energies(x, n, np.asarray(self.alphas), weight_db=w, as_db=True)[0, 0]
        return self.filters.flow, self.filters.fhigh, self.filters.f_nominal, db.astype(np.float64)
