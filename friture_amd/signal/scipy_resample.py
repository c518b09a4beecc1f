"""friture/signal/scipy_resample.py:51-141 on the GPU: Fourier-method resampling along axis 0, window=None (the only
form the hot path uses: Online_Linear_2D_resampler.set_height, friture/signal/online_linear_2D_resampler.py:45-55).
X = fft(x); the min(n, num) lowest frequencies are kept; y = ifft(Y) * num / n, real part for real input — two direct
DFT sums in float64 on the device (frt_fourier_resample), any lengths."""
from __future__ import annotations

import numpy as np

from .. import _lib


def resample(x, num, t=None, axis=0, window=None):
    if window is not None or t is not None:
        raise NotImplementedError("only the form the spectrogram uses: resample(x, num)")
    x = np.asarray(x)
    if np.iscomplexobj(x):
        raise NotImplementedError("real input only")
    x = np.moveaxis(np.asarray(x, np.float64), axis, -1)
    n = x.shape[-1]
    lead = x.shape[:-1]
    if min(n, int(num)) == 1:
        # the reference's second slice Y[-(N-1)//2:] is Y[0:] for N = 1: every bin receives X (scipy_resample.py:131-132)
        if n != 1:
            raise ValueError(f"could not broadcast input array from shape ({n},) into shape (1,)")      # as numpy raises there
        out = np.zeros(lead + (int(num),))
        out[..., 0] = x[..., 0] * int(num)
        return np.moveaxis(out, -1, axis)
    flat = np.ascontiguousarray(x.reshape(-1, n))
    out = np.empty((flat.shape[0], int(num)), np.float64)
    _lib.check(_lib.init().frt_fourier_resample(flat.ctypes.data, n, flat.shape[0], out.ctypes.data, int(num)))
    return np.moveaxis(out.reshape(lead + (int(num),)), -1, axis)
