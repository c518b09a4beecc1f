"""Design procedure of the octave filter bank coefficients (documentation + cross-check).

The reference ships its filter coefficients as generated artefacts (friture/generated_filters.py,
friture/data/generated_fft.npz) produced by friture/filter_design.py:131-205,364-435 and
friture/signal/lfilter.py:24-82.  This module re-derives the same designs with scipy and stores
them on demand (`design_all()`).  scipy's elliptic design changed between the version upstream
used and current releases, so the re-derived numbers agree with upstream's to 1e-5 .. 4e-3 only;
because every parity target is defined with upstream's numbers, the table the backend and the
oracle load (friture_amd/data/octave_filters.npz) holds those numbers verbatim, extracted by
tools/extract_reference_tables.py.  What is designed:

  * decimation low-pass: 12th-order elliptic, iirdesign(wp=0.48, ws=0.50, gpass=0.05, gstop=70)
  * band-passes of the top octave: ellip(2, 0.5 dB, 50 dB, [f_low, f_high]) for each of the
    `bpo` highest bands of the 9-octave series centred on 1 kHz (fi = 1000 * 2^(i/bpo))
  * 512-tap minimum-phase FIR equivalents (magnitude sampled on 8192 points, cepstral folding)
  * overlap-add FFT sizes per octave stage: smallest 5-smooth size >= 1024/2^j + 511

tests/test_filter_tables.py checks the committed table against the digest recorded next to the
reference (tests/golden/filter_tables.sha256) and keeps this re-derivation within tolerance.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

SAMPLING_RATE = 48000
NOCTAVE = 9
FIR_LENGTH = 512
BLOCK = 1024                     # largest input block of the FFT bank (octavefilters.py:30-35)
BANDS = (1, 3, 6, 12, 24)
DATA = Path(__file__).resolve().parent / "data" / "octave_filters.npz"


def octave_frequencies(total_bands: int, bands_per_octave: int):
    """Band centre and edge frequencies: fi = 1000 * 2^(i/bpo), edges fi * 2^(-+1/(2 bpo))."""
    half = total_bands // 2
    i = np.arange(-half, half) if total_bands % 2 == 0 else np.arange(-half, half + 1)
    step = 1.0 / bands_per_octave
    fi = 1000.0 * 2.0 ** (i * step)
    return fi, fi * np.sqrt(2.0 ** (-step)), fi * np.sqrt(2.0 ** step)


def design_decimator():
    from scipy.signal import iirdesign
    b, a = iirdesign(0.48, 0.50, 0.05, 70, analog=False, ftype="ellip", output="ba")
    return np.asarray(b, float), np.asarray(a, float)


def design_top_octave(bands_per_octave: int):
    from scipy.signal import ellip
    fi, flo, fhi = octave_frequencies(NOCTAVE * bands_per_octave, bands_per_octave)
    nyq = SAMPLING_RATE / 2.0
    B, A = [], []
    for lo, hi in zip(flo[-bands_per_octave:] / nyq, fhi[-bands_per_octave:] / nyq):
        hi = min(hi, 1.0)
        b, a = ellip(2, 0.5, 50, [lo, hi], btype="bandpass")
        B.append(b)
        A.append(a)
    return np.asarray(B, float), np.asarray(A, float)


def minimum_phase_fir(b, a, length: int = FIR_LENGTH, nfft: int = 8192):
    """FIR with the magnitude response of b/a and minimum phase (homomorphic / cepstral method)."""
    z = np.exp(-2j * np.pi * np.arange(nfft) / nfft)
    H = np.polyval(b[::-1], z) / np.polyval(a[::-1], z)
    ceps = np.fft.ifft(np.log(np.abs(H) + 1e-30)).real
    fold = np.zeros(nfft)
    fold[0] = ceps[0]
    fold[1:nfft // 2] = 2.0 * ceps[1:nfft // 2]
    fold[nfft // 2] = ceps[nfft // 2]
    h = np.fft.ifft(np.exp(np.fft.fft(fold))).real
    return h[:length]


def next_smooth_size(n: int) -> int:
    """Smallest size >= n whose prime factors are in {2, 3, 5}, but never above the next power of 2."""
    p2 = 1
    while p2 < n:
        p2 *= 2
    for size in range(n, p2):
        s = size
        for p in (2, 3, 5):
            while s % p == 0:
                s //= p
        if s == 1:
            return size
    return p2


def ola_fft_sizes():
    return [next_smooth_size(BLOCK // (2 ** j) + FIR_LENGTH - 1) for j in range(NOCTAVE)]


def design_all():
    out = {}
    bdec, adec = design_decimator()
    out["bdec"], out["adec"] = bdec, adec
    out["bdec_fir"] = minimum_phase_fir(bdec, adec)
    out["fft_sizes"] = np.asarray(ola_fft_sizes(), dtype=np.int64)
    for bpo in BANDS:
        B, A = design_top_octave(bpo)
        out[f"boct_{bpo}"], out[f"aoct_{bpo}"] = B, A
        out[f"boct_fir_{bpo}"] = np.asarray([minimum_phase_fir(b, a) for b, a in zip(B, A)])
    return out


def load_tables() -> dict:
    with np.load(DATA) as z:
        return {k: z[k] for k in z.files}


def fir_responses(fir: np.ndarray, fft_sizes) -> list[np.ndarray]:
    """rfft of zero-padded FIR taps at each stage size (what filter_design.py:399-414 tabulates)."""
    fir = np.atleast_2d(fir)
    return [np.fft.rfft(fir, int(n), axis=-1) for n in fft_sizes]


if __name__ == "__main__":
    shipped, mine = load_tables(), design_all()
    for key in sorted(mine):
        err = np.max(np.abs(mine[key] - shipped[key])) / np.max(np.abs(shipped[key]))
        print(f"{key:14s} re-derived vs shipped: {err:.3e}")
