"""One-off host tables of the hot path (window, frequency axis, psychoacoustic weightings).

These are the quantities the reference caches per FFT size (friture/audioproc.py:73-96) and per
band layout (friture/octavefilters.py:76-82).  They are computed once per configuration, in
float64 on the host, and handed to the kernels as plan constants; widgets also read them as
numpy arrays, so they stay host-visible.
"""
from __future__ import annotations

import numpy as np

from .constants import SAMPLING_RATE


def hann_symmetric(n: int) -> np.ndarray:
    """w[k] = (1 - cos(2 pi k / (n - 1))) / 2, the end-point-inclusive Hann window."""
    k = np.arange(n, dtype=np.float64)
    return 0.5 * (1.0 - np.cos(2.0 * np.pi * k / (n - 1)))


def rfft_frequencies(n: int) -> np.ndarray:
    """Bin centres of an n-point real FFT at the stream rate: n/2 + 1 points on [0, fs/2]."""
    return np.linspace(0, SAMPLING_RATE // 2, n // 2 + 1)


def weighting_db(f, floor: float = 0.0):
    """IEC 61672 A, B and C frequency weightings in dB at frequencies `f`.

    R_C = c^2 f^2 / ((f^2 + a^2)(f^2 + c^2)) with a = 20.6 Hz, c = 12200 Hz;
    R_B = R_C f / sqrt(f^2 + 158.5^2);  R_A = R_C f^2 / sqrt((f^2 + 107.7^2)(f^2 + 737.9^2)),
    evaluated in the reference's operation order so that the tables are bit-identical; offsets
    +2.0, +0.17, +0.06 dB.  `floor` is added inside the logarithm (1e-50 on FFT bins, 0 on bands).
    """
    f = np.asarray(f, np.float64)
    pole = (f ** 2 + 20.6 ** 2) * (f ** 2 + 12200.0 ** 2)
    rc = 12200.0 ** 2 * f ** 2 / pole
    rb = 12200.0 ** 2 * f ** 3 / (pole * ((f ** 2 + 158.5 ** 2) ** 0.5))
    ra = 12200.0 ** 2 * f ** 4 / (pole * ((f ** 2 + 107.7 ** 2) ** 0.5) * ((f ** 2 + 737.9 ** 2) ** 0.5))
    with np.errstate(divide="ignore"):
        return (2.0 + 20.0 * np.log10(ra + floor), 0.17 + 20.0 * np.log10(rb + floor),
                0.06 + 20.0 * np.log10(rc + floor))
