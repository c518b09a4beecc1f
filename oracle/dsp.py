"""oracle/dsp.py — CPU restatement of Friture's spectral-analysis hot path (TEST INFRASTRUCTURE).

This module is the checker the HIP backend is compared against.  It is *not* part of the product:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it, and nothing in
friture_amd/ does.  Arithmetic is float64 numpy, the same as the reference; every function cites
the reference lines it restates (paths relative to the reference checkout).

Parity pin: upstream has no golden vectors for most of this path, so the
oracle is pinned against the *reference itself*, executed unmodified in the build container
through oracle/refshim.py; oracle/make_golden.py records those outputs in tests/golden/*.npz
and tests/test_oracle_golden.py replays them wherever the tests run (including the GPU box,
where the reference checkout does not exist).  The upstream property tests that do exist
(friture/test/test_octave_filters.py:37-100 energy ±5 %, decimation ordering :63-72,
friture/test/test_exp_smoothing.py) are replayed against the oracle in the same test file.
"""
from __future__ import annotations

import ctypes
from pathlib import Path

import numpy as np

SAMPLING_RATE = 48000            # friture/audiobackend.py:31
FRAMES_PER_BUFFER = 512          # friture/audiobackend.py:32
NOCTAVE = 9                      # friture/filter.py:7
FIR_LENGTH = 512                 # friture/octavefilters.py:35

_HERE = Path(__file__).resolve().parent
_TABLES = _HERE.parent / "friture_amd" / "data" / "octave_filters.npz"


# --------------------------------------------------------------------------------------------
# P1 / P2: windowed rFFT power spectrum and its tables
# --------------------------------------------------------------------------------------------

def hann_symmetric(n_fft: int) -> np.ndarray:
    """Symmetric Hann window 0.5 (1 - cos(2 pi n / (N-1)))  — friture/audioproc.py:76-80."""
    n = np.arange(0, n_fft)
    return 0.5 * (1.0 - np.cos(2 * np.pi * n / (n_fft - 1)))


def psd_frame(samples: np.ndarray, window: np.ndarray) -> np.ndarray:
    """|rfft(x w)|^2 / N^2  — friture/audioproc.py:42-50 (analyzelive + norm_square)."""
    spec = np.fft.rfft(samples * window)
    return (spec * spec.conjugate()).real / float(len(window)) ** 2


def frequency_axis(n_fft: int) -> np.ndarray:
    """linspace(0, fs//2, N/2+1)  — friture/audioproc.py:86."""
    return np.linspace(0, SAMPLING_RATE // 2, n_fft // 2 + 1)


def _abc_ratios(f: np.ndarray):
    f2 = f ** 2
    common = (f2 + 20.6 ** 2) * (f2 + 12200.0 ** 2)
    rc = 12200.0 ** 2 * f2 / common
    rb = 12200.0 ** 2 * f ** 3 / (common * ((f2 + 158.5 ** 2) ** 0.5))
    ra = 12200.0 ** 2 * f ** 4 / (common * ((f2 + 107.7 ** 2) ** 0.5) * ((f2 + 737.9 ** 2) ** 0.5))
    return ra, rb, rc


def weighting_curves(f: np.ndarray):
    """A, B, C weighting in dB on the FFT bins (eps = 1e-50)  — friture/audioproc.py:88-96."""
    ra, rb, rc = _abc_ratios(np.asarray(f, float))
    eps = 1e-50
    return 2.0 + 20.0 * np.log10(ra + eps), 0.17 + 20.0 * np.log10(rb + eps), 0.06 + 20.0 * np.log10(rc + eps)


def band_weighting(fi: np.ndarray):
    """A, B, C weighting at band centres (no eps)  — friture/octavefilters.py:76-82."""
    with np.errstate(divide="ignore"):
        ra, rb, rc = _abc_ratios(np.asarray(fi, float))
        return 2.0 + 20.0 * np.log10(ra), 0.17 + 20.0 * np.log10(rb), 0.06 + 20.0 * np.log10(rc)


def stft_psd(x: np.ndarray, n_fft: int, hop: int) -> np.ndarray:
    """Frame loop of the STFT drivers, frame-major result [F, N/2+1].

    friture/spectrum.py:144-155 and friture/spectrogram.py:149-159 take the window ending at
    old_index and advance by int(N (1 - overlap)); on a batch of T samples that is frame f =
    x[f*hop : f*hop + N] for f < (T-N)//hop + 1.  (The reference stores the transpose.)
    """
    x = np.asarray(x, np.float64)
    w = hann_symmetric(n_fft)
    n_frames = (len(x) - n_fft) // hop + 1 if len(x) >= n_fft else 0
    out = np.empty((max(n_frames, 0), n_fft // 2 + 1))
    for f in range(n_frames):
        out[f] = psd_frame(x[f * hop: f * hop + n_fft], w)
    return out


# --------------------------------------------------------------------------------------------
# P4 / P7: dB, normalisation, colour look-up
# --------------------------------------------------------------------------------------------

def log_spectrum(psd: np.ndarray) -> np.ndarray:
    """10 log10(P + 1e-30)  — friture/spectrogram.py:119-125, friture/spectrum.py:95-101."""
    return 10.0 * np.log10(psd + 1e-30)


def normalise(db: np.ndarray, spec_min: float, spec_max: float) -> np.ndarray:
    """(dB - min)/(max - min), unclipped  — friture/spectrogram.py:128-129."""
    return (db - spec_min) / (spec_max - spec_min)


def cmrmap(n: int = 256) -> np.ndarray:
    """Monochrome-compatible colour map: cubic splines through 9 anchors, rescaled to [0, 1]
    — friture/plotting/cmrmap_generate.py:57-83 (values shipped as generated_cmrmap.py)."""
    from scipy.interpolate import splev, splrep
    anchors = np.array([[0, 0, 0], [0.1, 0.1, 0.35], [0.3, 0.15, 0.65], [0.6, 0.2, 0.50], [1, 0.25, 0.15],
                        [0.9, 0.55, 0], [0.9, 0.75, 0.1], [0.9, 0.9, 0.5], [1, 1, 1]], float)
    xr = np.linspace(0, 1, anchors.shape[0])
    x = np.linspace(0, 1, n)
    cm = np.stack([splev(x, splrep(xr, anchors[:, i], s=0)) for i in range(3)], axis=1)
    cm -= np.min(cm)
    cm /= np.max(cm)
    return cm


def colour_lut(cmap: np.ndarray) -> np.ndarray:
    """0xFFRRGGBB words from int(c*255) truncation (QColor(r,g,b).rgb())
    — friture/signal/color_tranform.py:36-46."""
    q = (np.asarray(cmap) * 255).astype(np.int64)   # int() truncation of non-negative values
    return (0xFF000000 | (q[:, 0] << 16) | (q[:, 1] << 8) | q[:, 2]).astype(np.uint32)


def colour_pixels(lut: np.ndarray, values: np.ndarray) -> np.ndarray:
    """lut[int(clip(v, 0, 1) * 255)]  — friture/signal/color_tranform.py:48-51,
    friture/signal/lookup_table.py:50-52."""
    v = np.clip(values, 0.0, 1.0)
    return lut[(v * 255).astype(np.intp)]


def colour_index(psd, weight_db, spec_min, spec_max):
    """The reference's float64 epilogue applied to a given power spectrum: returns (idx, v255) with
    v255 = clip((10 log10(P + 1e-30) + w - min)/(max - min), 0, 1) * 255 and idx = int(v255)
    — friture/spectrogram.py:119-129,161-162, friture/signal/color_tranform.py:48-51,
    friture/signal/lookup_table.py:50-52."""
    db = log_spectrum(np.asarray(psd, np.float64))
    if weight_db is not None:
        db = db + np.asarray(weight_db)[None, :]
    v255 = np.clip(normalise(db, spec_min, spec_max), 0.0, 1.0) * 255
    return v255.astype(np.intp), v255


def image_parity(image, psd_same_path, psd_reference, weight_db, spec_min, spec_max, lut, edge=1e-6):
    """Parity accounting of a colour image (pixel-exact except where v*255 is within
    `edge` of an integer).  `psd_same_path` is the power spectrum the image was derived from (the
    float32 PSD of the same frames), `psd_reference` the reference's float64 PSD or None.  Separates
    the two sources of a differing pixel: the epilogue itself (dB -> normalise -> index -> LUT, which
    must be exact given its input) and the float32 transform's error in P moving a bin across an edge."""
    lut = np.asarray(lut)
    image = np.asarray(image).view(np.uint32)
    idx, v255 = colour_index(psd_same_path, weight_db, spec_min, spec_max)
    frac = v255 - np.floor(v255)
    near = (frac < edge) | (frac > 1.0 - edge)
    bad = image != lut[idx]
    out = {"pixels_checked": int(image.size),
           "epilogue_mismatched": int(np.sum(bad)),
           "epilogue_mismatch_outside_edge": int(np.sum(bad & ~near)),
           "edge": edge}
    if psd_reference is not None:
        ridx, r255 = colour_index(psd_reference, weight_db, spec_min, spec_max)
        rfrac = r255 - np.floor(r255)
        rnear = (rfrac < edge) | (rfrac > 1.0 - edge)
        rbad = image != lut[ridx]
        out["pixels_mismatched"] = int(np.sum(rbad))
        out["mismatch_outside_edge"] = int(np.sum(rbad & ~rnear))
        # how far the float32 power moved the index value of the differing pixels
        out["max_index_shift_of_mismatch"] = float(np.max(np.abs(v255 - r255)[rbad])) if np.any(rbad) else 0.0
        ref = np.asarray(psd_reference, np.float64)
        err_frame = np.max(np.abs(np.asarray(psd_same_path, np.float64) - ref), axis=-1, keepdims=True)   # e_f: every bin's |dP| <= e_f
        out["psd_rel_max"] = float(np.max(err_frame[..., 0] / np.max(ref, axis=-1)))
        # Accounting of every differing pixel: the index value is q = 255 (10 log10(P + 1e-30) + w - min) / (max - min), so a
        # power error of at most e_f moves it by at most g |ln((P + 1e-30 -+ e_f) / (P + 1e-30))|, g = 2550 / (ln 10 (max - min))
        # (unbounded when e_f reaches P: a bin at the error floor may take any colour below its own).  A differing pixel
        # is ACCOUNTED FOR when the reference's q lies within that distance of the integer edge it was carried across;
        # anything else is a defect of the transform or the epilogue, whatever its count.
        g = 2550.0 / (np.log(10.0) * abs(spec_max - spec_min))
        p = ref + 1e-30
        with np.errstate(divide="ignore", invalid="ignore"):
            up = g * np.log1p(err_frame / p)
            down = np.where(err_frame < p, -g * np.log1p(-np.minimum(err_frame / p, 1.0 - 1e-16)), np.inf)
        reach = np.maximum(up, down) + 1e-9                       # + the float64 rounding of the reference's own expression
        # distance of the reference's UNCLIPPED index value from the nearest edge: the index changes at q = 1, 2, ..., 255
        dbr = log_spectrum(ref)
        if weight_db is not None:
            dbr = dbr + np.asarray(weight_db)[None, :]
        q = 255.0 * normalise(dbr, spec_min, spec_max)
        to_edge = np.abs(q - np.clip(np.rint(q), 1.0, 255.0))
        out["mismatch_unaccounted"] = int(np.sum(rbad & ~(to_edge <= reach)))
        out["max_reach_of_mismatch"] = float(np.max(reach[rbad])) if np.any(rbad) else 0.0
    return out


def spectrogram_image(x, n_fft, hop, weight_db, spec_min, spec_max, lut):
    """STFT -> dB + weighting -> normalise -> colour, without the screen-space resamplers
    (friture/spectrogram.py:147-162 followed directly by Color_Transform.push)."""
    db = log_spectrum(stft_psd(x, n_fft, hop))
    if weight_db is not None:
        db = db + np.asarray(weight_db)[None, :]
    return colour_pixels(lut, normalise(db, spec_min, spec_max))


# --------------------------------------------------------------------------------------------
# P8: exponential smoothing over a block
# --------------------------------------------------------------------------------------------

def smoothing_kernel(alpha: float, n: int) -> np.ndarray:
    """(1-alpha)^(N-1 .. 0)  — friture/octavespectrum.py:77-81, friture/spectrum.py:196-222."""
    return (1.0 - alpha) ** np.arange(n - 1, -1, -1)


def exp_smoothed_value(kernel, alpha, data, previous):
    """alpha * dot(kernel[-N:], data[:N]) + previous (1-alpha)^N  — friture/signal/exp_smoothing.py:40-56."""
    n, nk = data.shape[0], kernel.shape[0]
    if n > nk:
        n, decay = nk, 0.0
    else:
        decay = (1.0 - alpha) ** n
    if n == 0:
        return previous
    return float(alpha * np.dot(kernel[nk - n:nk], data[:n]) + previous * decay)


def exp_smoothed_value_2d(kernel, alpha, data, previous):
    """Row-wise version  — friture/signal/exp_smoothing.py:91-107."""
    nt, nk = data.shape[1], kernel.shape[0]
    if nt > nk:
        nt, decay = nk, 0.0
    else:
        decay = (1.0 - alpha) ** nt
    if nt == 0:
        return np.array(previous, copy=True)
    return alpha * (data[:, :nt] @ kernel[nk - nt:nk]) + previous * decay


def harmonic_product_spectrum(sp):
    """sp[:K] * sp[::2][:K] * sp[::3][:K] with K = len(sp) // 3  — friture/spectrum.py:103-123."""
    k = sp.shape[0] // 3
    return sp[:k] * sp[::2][:k] * sp[::3][:k]


def spectrum_readout(spn, kernel, alpha, previous, weight, freq, ref_smoothed=None):
    """Post-processing of Spectrum_Widget.handle_new_data (friture/spectrum.py:156-182) for one channel.
    spn: (bins, frames).  Returns dict(smoothed, db, peak_index, pitch_index, fmax, fpitch)."""
    sp = exp_smoothed_value_2d(kernel, alpha, spn, previous)
    if ref_smoothed is not None:
        db = log_spectrum(sp) - log_spectrum(ref_smoothed)
    else:
        db = log_spectrum(sp) + weight
    i = int(np.argmax(db))
    hps = harmonic_product_spectrum(ref_smoothed if ref_smoothed is not None else sp)
    p = int(np.argmax(hps))
    return dict(smoothed=sp, db=db, peak_index=i, pitch_index=p, fmax=freq[i], fpitch=max(freq[p], 1e-20))


# --------------------------------------------------------------------------------------------
# O1 / G2: direct-form-II-transposed IIR, decimation, exact octave bank
# --------------------------------------------------------------------------------------------

_iir_c = None


def _load_iir_c():
    """oracle/iir_ref.c compiled by `make -C oracle` (optional speed-up, same IEEE operations)."""
    global _iir_c
    if _iir_c is None:
        so = _HERE / "_build" / "libiir_ref.so"
        if not so.exists():                      # not built yet (fresh checkout): one quiet attempt with the Makefile
            import subprocess
            try:
                subprocess.run(["make", "-C", str(_HERE)], capture_output=True, timeout=120, check=False)
            except Exception:
                pass
        if so.exists():
            lib = ctypes.CDLL(str(so))
            lib.oracle_lfilter_df2t.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p, ctypes.c_long,
                                                                                          ctypes.c_void_p, ctypes.c_void_p]
            lib.oracle_lfilter_df2t.restype = None
            _iir_c = lib
        else:
            _iir_c = False
    return _iir_c


def lfilter_df2t(b, a, x, zi, force_python: bool = False):
    """Direct form II transposed with carried state; a[0] == 1 is assumed as in the reference.

        y[k]   = z[0] + b[0] x[k]
        z[n]   = z[n+1] + x[k] b[n+1] - y[k] a[n+1]      (n < order-1)
        z[-1]  = x[k] b[-1] - y[k] a[-1]
    — friture/signal/lfilter.py:131-139.  Returns (y, zf).
    """
    b = np.ascontiguousarray(b, np.float64)
    a = np.ascontiguousarray(a, np.float64)
    x = np.ascontiguousarray(x, np.float64)
    z = np.array(zi, np.float64, copy=True)
    assert b.shape == a.shape and z.shape[0] == b.shape[0] - 1
    y = np.empty_like(x)
    lib = None if force_python else _load_iir_c()
    if lib:
        lib.oracle_lfilter_df2t(b.ctypes.data, a.ctypes.data, len(b), x.ctypes.data, len(x), z.ctypes.data, y.ctypes.data)
        return y, z
    bl, al, zl = b.tolist(), a.tolist(), z.tolist()
    nb = len(bl)
    for k, xk in enumerate(x.tolist()):
        yk = zl[0] + bl[0] * xk if nb > 1 else xk * bl[0]
        y[k] = yk
        for n in range(nb - 2):
            zl[n] = zl[n + 1] + xk * bl[n + 1] - yk * al[n + 1]
        if nb > 1:
            zl[nb - 2] = xk * bl[nb - 1] - yk * al[nb - 1]
    return y, np.array(zl)


def decimate(bdec, adec, x, zi=None):
    """Low-pass then keep every other sample starting at 0  — friture/signal/decimate.py:27-42."""
    if len(x) == 0:
        raise Exception("Filter input is too small")
    if zi is None:
        zi = np.zeros(max(len(bdec), len(adec)) - 1)
    y, zf = lfilter_df2t(bdec, adec, x, zi)
    return y[::2], zf


def decimate_multiple(ndec, bdec, adec, x, zis):
    """Chain of `ndec` decimations with per-stage state  — friture/signal/decimate.py:45-71."""
    if x.size == 0:
        return x, zis
    out = x
    if zis is None:
        for _ in range(ndec):
            out, _zf = decimate(bdec, adec, out)
        return out, None
    zfs = []
    for _, zi in zip(range(ndec), zis):
        out, zf = decimate(bdec, adec, out, zi)
        zfs.append(zf)
    return out, zfs


def decimate_multiple_filtic(ndec, bdec, adec):
    """Zero initial conditions  — friture/signal/decimate.py:74-84."""
    return [np.zeros(max(len(bdec), len(adec)) - 1) for _ in range(ndec)]


def iir_bank_filtic(bdec, adec, boct, aoct):
    """Zero states in processing order  — friture/filter.py:121-133."""
    zs = []
    for _ in range(NOCTAVE):
        for i in range(len(boct))[::-1]:
            zs.append(np.zeros(max(len(boct[i]), len(aoct[i])) - 1))
        zs.append(np.zeros(max(len(bdec), len(adec)) - 1))
    return zs


def iir_bank(bdec, adec, boct, aoct, x, zis):
    """Exact IIR octave bank with decimation (legacy path)  — friture/filter.py:86-118.

    Octave j filters the j-times decimated signal with the top-octave band-passes (highest band
    first), so band k = 9*bpo-1 .. 0 is filled from the top; dec[k] = 2^j.
    """
    bpo = len(boct)
    nb = NOCTAVE * bpo
    y = [None] * nb
    dec = [0] * nb
    zfs = []
    m, k, sig = 0, nb - 1, x
    for j in range(NOCTAVE):
        for i in range(bpo)[::-1]:
            y[k], zf = lfilter_df2t(boct[i], aoct[i], sig, zis[m])
            zfs.append(zf)
            dec[k] = 2 ** j
            m += 1
            k -= 1
        sig, zf = decimate(bdec, adec, sig, zis[m])
        zfs.append(zf)
        m += 1
    return y, dec, zfs


# --------------------------------------------------------------------------------------------
# O2 / O3: production FFT overlap-add bank and band tables
# --------------------------------------------------------------------------------------------

def octave_frequencies(total_bands, bands_per_octave):
    """fi = 1000 2^(i/bpo); edges fi 2^(-+1/(2 bpo))  — friture/filter.py:39-54."""
    half = total_bands // 2
    i = np.arange(-half, half) if total_bands % 2 == 0 else np.arange(-half, half + 1)
    b = 1.0 / bands_per_octave
    fi = 1000.0 * 2 ** (i * b)
    return fi, fi * np.sqrt(2 ** (-b)), fi * np.sqrt(2 ** b)


def get_decs(bands_per_octave):
    """Decimation factor of each band, low band first  — friture/octavefilters.py:60-63."""
    return [2 ** j for j in range(NOCTAVE)[::-1] for _ in range(bands_per_octave)]


def load_filter_tables() -> dict:
    with np.load(_TABLES) as z:
        return {k: z[k] for k in z.files}


class OlaBank:
    """Octave_Filters.filter: every IIR replaced by its 512-tap minimum-phase FIR, applied by FFT
    overlap-add at sizes [1536, 1024, 768, 640, 576, 576, 540, 540, 540] with a pending tail of
    511 samples per stage  — friture/octavefilters.py:49-58,123-158, friture/filter.py:136-247."""

    def __init__(self, bands_per_octave: int, tables: dict | None = None):
        t = tables or load_filter_tables()
        self.bpo = bands_per_octave
        self.nbands = NOCTAVE * bands_per_octave
        self.fft_sizes = [int(s) for s in t["fft_sizes"]]
        fir_oct = t[f"boct_fir_{bands_per_octave}"]
        fir_dec = t["bdec_fir"]
        # frequency responses: rfft of the zero-padded taps (filter_design.py:399-414)
        self.H_oct = [np.fft.rfft(fir_oct, n, axis=1) for n in self.fft_sizes]
        self.H_dec = [np.fft.rfft(fir_dec, n) for n in self.fft_sizes]
        self.reset()

    def reset(self):
        self.tail_oct = [np.zeros((self.bpo, FIR_LENGTH - 1)) for _ in range(NOCTAVE)]
        self.tail_dec = [np.zeros(FIR_LENGTH - 1) for _ in range(NOCTAVE)]

    def filter(self, x):
        lm1 = FIR_LENGTH - 1
        y = [None] * self.nbands
        dec = [0] * self.nbands
        k = self.nbands - 1
        sig = np.asarray(x, np.float64)
        for j in range(NOCTAVE):
            ns = len(sig)
            nfft = self.fft_sizes[j]
            X = np.fft.rfft(sig, nfft)                       # crops when ns > nfft (filter.py:206)
            full_oct = np.fft.irfft(X[None, :] * self.H_oct[j], nfft, axis=1)
            full_dec = np.fft.irfft(X * self.H_dec[j], nfft)
            pend_o, pend_d = self.tail_oct[j], self.tail_dec[j]
            n_add = min(lm1, ns)
            if n_add > 0:
                full_oct[:, :n_add] += pend_o[:, :n_add]
                full_dec[:n_add] += pend_d[:n_add]
            for i in range(self.bpo)[::-1]:
                y[k] = full_oct[i, :ns]
                dec[k] = 2 ** j
                k -= 1
            sig = full_dec[:ns:2]
            new_o = full_oct[:, ns:ns + lm1].copy()
            rest = pend_o[:, n_add:]
            if rest.shape[1] > 0:
                new_o[:, :rest.shape[1]] += rest
            self.tail_oct[j] = new_o
            new_d = full_dec[ns:ns + lm1].copy()
            rest_d = pend_d[n_add:]
            if len(rest_d) > 0:
                new_d[:len(rest_d)] += rest_d
            self.tail_dec[j] = new_d
        return y, dec


# --------------------------------------------------------------------------------------------
# O4: band energies of the octave-spectrum widget
# --------------------------------------------------------------------------------------------

def band_smoothing_setup(bands_per_octave: int, response_time: float):
    """alphas and kernels per band  — friture/octavespectrum.py:140-156."""
    w = 0.65
    decs = get_decs(bands_per_octave)
    ns = [response_time * SAMPLING_RATE / d for d in decs]
    lens = [2 * 4096 / d for d in decs]
    alphas = [1.0 - (1.0 - w) ** (1.0 / (n + 1)) for n in ns]
    kernels = [smoothing_kernel(a, int(n)) for a, n in zip(alphas, lens)]
    return alphas, kernels


def band_energies(y, kernels, alphas, previous):
    """sp[b] = exp_smoothed_value(kernel_b, alpha_b, y_b^2, old_b)  — friture/octavespectrum.py:104."""
    return [exp_smoothed_value(k, a, band ** 2, old) for band, k, a, old in zip(y, kernels, alphas, previous)]


def band_db(sp, weight=0.0):
    """10 log10(sp + 1e-30) + w  — friture/octavespectrum.py:120-121."""
    return 10 * np.log10(np.asarray(sp) + 1e-30) + weight


# --------------------------------------------------------------------------------------------
# G1 / G3: GCC-PHAT and the delay read-out
# --------------------------------------------------------------------------------------------

def gcc_phat(d0, d1):
    """Generalised cross-correlation with phase transform  — friture/signal/correlation.py:24-43.

    The reference subtracts the means *in place* on its arguments; this restatement works on
    copies and returns (xcorr, d0_demeaned, d1_demeaned) so callers can reproduce that effect.
    """
    d0 = np.array(d0, np.float64, copy=True)
    d1 = np.array(d1, np.float64, copy=True)
    d0 -= d0.mean()
    d1 -= d1.mean()
    win = np.hanning(len(d0))
    D0 = np.fft.rfft(d0 * win)
    D1 = np.fft.rfft(d1 * win)
    G = D0.conjugate() * D1
    mag = np.abs(G)
    W = 1.0 / (1e-10 * max(mag) + mag)
    return np.fft.irfft(W * G), d0, d1


def delay_readout(xcorr, old_xcorr, subsampled_rate, delayrange_s):
    """Smoothing, peak pick, delay / polarity / confidence  — friture/delay_estimator.py:134-176.

    Returns dict(delay_ms, extremum, correlation_pct, distance_m, smoothed)."""
    if old_xcorr is not None and old_xcorr.shape == xcorr.shape:
        sm = 0.3 * xcorr + (1.0 - 0.3) * old_xcorr
    else:
        sm = xcorr
    i = int(np.argmax(np.abs(sm)))
    peak_norm = abs(sm[i]) / (3 * np.std(sm))
    t = 2 * delayrange_s
    delay_ms = 1e3 * float(i) / subsampled_rate
    if delay_ms > 1e3 * t / 2.0:
        delay_ms -= 1e3 * t
    xx = (peak_norm > 1.0) * (peak_norm - 1.0)
    xx = (0.12 * xx) ** 3
    return dict(delay_ms=delay_ms, extremum=float(sm[i]), correlation_pct=int((xx / (1.0 + xx)) * 100),
                distance_m=delay_ms * 1e-3 * 340.0, smoothed=sm, argmax=i)


# --------------------------------------------------------------------------------------------
# R1: mirror ring buffer
# --------------------------------------------------------------------------------------------

class MirrorRing:
    """Double-length mirrored ring: every push is written twice so that any window is one
    contiguous slice; grows by x1.5 on demand  — friture/ringbuffer.py:28-130."""

    def __init__(self, length: int = 10000):
        self.buffer_length = length
        self.buffer = np.zeros((1, 2 * length))
        self.offset = 0

    def push(self, data: np.ndarray):
        dim, n = data.shape
        if dim != self.buffer.shape[0]:
            self.buffer = np.zeros((dim, 2 * self.buffer_length))
        self._grow(n)
        L = self.buffer_length
        o = self.offset % L
        self.buffer[:, o:o + n] = data
        direct = min(n, L - o)
        self.buffer[:, o + L:o + L + direct] = data[:, :direct]
        self.buffer[:, :n - direct] = data[:, direct:]
        self.offset += n

    def data(self, length: int):
        self._grow(length)
        stop = self.offset % self.buffer_length + self.buffer_length
        return self.buffer[:, stop - length:stop]

    def data_older(self, length: int, delay: int):
        self._grow(length + delay)
        start = (self.offset - length - delay) % self.buffer_length + self.buffer_length
        return self.buffer[:, start:start + length]

    def data_indexed(self, start: int, length: int):
        """The `length` samples ending at absolute index `start`  — ringbuffer.py:87-99."""
        self._grow(length + self.offset - start)
        stop0 = start % self.buffer_length + self.buffer_length
        start0 = stop0 - length
        if start0 < 0 or start0 > 2 * self.buffer_length:
            raise ArithmeticError("Start index is wrong %d %d" % (start0, self.buffer_length))
        return self.buffer[:, start0:stop0]

    def _grow(self, length: int):
        if length <= self.buffer_length:
            return
        old, new = self.buffer_length, int(1.5 * length)
        nb = np.zeros((self.buffer.shape[0], 2 * new))
        shift = (self.offset % new - self.offset % old) % new
        nb[:, shift:shift + old] = self.buffer[:, :old]
        direct = min(old, new - shift)
        nb[:, new + shift:new + shift + direct] = self.buffer[:, :direct]
        nb[:, :old - direct] = self.buffer[:, direct:old]
        self.buffer, self.buffer_length = nb, new


# --------------------------------------------------------------------------------------------
# P5 / P6: screen-space resamplers of the spectrogram
# --------------------------------------------------------------------------------------------

class Scale:
    """Frequency scales: transform / inverse  — friture/plotting/frequency_scales.py:65-293."""
    ERB_A = 21.33228113095401739888262
    FWD = {
        "linear": lambda f: f,
        "log": lambda f: np.log10(f),
        "mel": lambda f: 2595 * np.log10(1 + f / 700),
        "erb": lambda f: Scale.ERB_A * np.log10(1 + 0.00437 * f),
        "octave": lambda f: np.log2(np.fmax(f, 1e-20)),
    }
    INV = {
        "linear": lambda v: v,
        "log": lambda v: 10 ** v,
        "mel": lambda v: 700 * (10 ** (v / 2595) - 1),
        "erb": lambda v: (10 ** (v / Scale.ERB_A) - 1) / 0.00437,
        "octave": lambda v: 2 ** v,
    }


def frequency_targets(scale: str, fmin: float, fmax: float, n: int) -> np.ndarray:
    """scale.inverse(linspace(T(fmin), T(fmax), n))  — friture/signal/frequency_resampler.py:44-49."""
    return Scale.INV[scale](np.linspace(Scale.FWD[scale](fmin), Scale.FWD[scale](fmax), n))


def frequency_resample(targets, freq, data):
    """Per column np.interp(targets, freq, column); data is [bins, columns]
    — friture/signal/frequency_resampler.py:67-83."""
    out = np.zeros((targets.size, data.shape[1]))
    for j in range(data.shape[1]):
        out[:, j] = np.interp(targets, freq, data[:, j])
    return out


def fourier_resample(x, num):
    """Fourier-method resampling along axis 0 (window None): the min(n, num) lowest frequencies of fft(x) zero-padded /
    truncated to num bins, y = ifft(Y) * num / n, real part  — friture/signal/scipy_resample.py:108-141."""
    x = np.asarray(x, np.float64)
    X = np.fft.fft(x, axis=0)
    nx = x.shape[0]
    n = int(min(num, nx))
    Y = np.zeros((num,) + x.shape[1:], complex)
    Y[0:(n + 1) // 2] = X[0:(n + 1) // 2]
    Y[-(n - 1) // 2:] = X[-(n - 1) // 2:]          # floor division of the negative number, as the reference writes it
    return (np.fft.ifft(Y, axis=0) * (float(num) / float(nx))).real


class TimeResampler:
    """Stateful linear resampling along time to the pixel rate (ratio = L/M)
    — friture/signal/online_linear_2D_resampler.py:14-97, friture/signal/linear_interp.py:11-62.
    A push of another height Fourier-resamples the carried column first (set_height, :45-55)."""

    def __init__(self, L, M, height):
        self.ratio = float(L) / M
        self.height = height
        self.orig_index = 0.0
        self.resampled_index = 0.0
        self.old = np.zeros(height)

    def _processable(self, m):
        return int(np.ceil((self.orig_index + m - (self.resampled_index + self.ratio)) / self.ratio))

    def push(self, data):
        if data.shape[0] != self.height:
            self.height = data.shape[0]
            self.orig_index = 0.0
            self.resampled_index = 0.0
            self.old = fourier_resample(self.old, self.height)
        cols = data.shape[1]
        out = np.zeros((self.height, self._processable(cols)))
        w = 0
        for j in range(cols):
            self.orig_index += 1.0
            n = self._processable(0)
            if n > 0:
                idx = self.resampled_index + self.ratio * np.arange(1, n + 1, dtype=np.float64)
                a = self.orig_index - idx
                out[:, w:w + n] = data[:, j][:, None] * (1.0 - a)[None, :] + self.old[:, None] * a[None, :]
                self.resampled_index = float(idx[-1])
                w += n
            self.old = data[:, j]
        return out


# --------------------------------------------------------------------------------------------
# T1: SWIPE-style pitch tracker
# --------------------------------------------------------------------------------------------
# Parity pin: friture/test/test_pitch_tracker.py:42-71 holds two known answers (3000 and 1500),
# but the reference's current estimate_pitch does not reproduce them (executed here through
# oracle/refshim.py it returns nan for both: the candidates stop at max_freq = 1047 Hz and the
# confidence gate rejects the 32-point frames).  The pin is therefore the reference code itself,
# executed unmodified in the build container: oracle/make_golden_pitch.py -> tests/golden/pitch.npz.

SWIPE_HARMONICS = (1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 13, 17, 19, 23)     # friture/pitch_tracker.py:217


def parabolic_vertex(y1, y2, y3):
    """Vertex offset and height of the parabola through (-1,y1), (0,y2), (1,y3)
    — friture/pitch_tracker.py:160-193 (eps in the denominator included)."""
    a = (y1 - 2 * y2 + y3) / 2
    b = (y3 - y1) / 2
    vx = -b / (2 * a + np.finfo(np.float64).eps)
    return vx, a * vx ** 2 + b * vx + y2


def swipe_kernel(f: float, freqs: np.ndarray) -> np.ndarray:
    """One candidate's kernel over the log-spaced grid — friture/pitch_tracker.py:195-264.
    Cosine peak lobes (half-width 0.15 of the harmonic spacing) at the selected harmonics, quarter
    height at the others, negative half-height valley lobes between; sqrt(1/f) decay above harmonic
    2.15; positive area normalised to 1; divided by the fraction of harmonics below Nyquist."""
    harmonics = np.array(SWIPE_HARMONICS)
    peak_w = 0.15
    valley_w = 1 - peak_w
    n_possible = min(int(freqs[-1] / f), len(harmonics))
    selected = harmonics[:n_possible]
    ratio = freqs / f
    k = np.zeros_like(freqs)
    for i in np.arange(1, harmonics[-1] + 1):
        a = ratio - i
        chosen = i in selected
        valley = np.logical_and(-valley_w < a, a < (-peak_w if chosen else peak_w))
        k[valley] = -np.cos((a[valley] + 0.5) / ((valley_w - peak_w) / 2) * (np.pi / 2)) / 2
        peak = np.abs(a) < peak_w
        k[peak] = np.cos(a[peak] / peak_w * (np.pi / 2)) / (1 if chosen else 4)
    knee = f * (2 + peak_w)
    k *= np.where(freqs <= knee, np.sqrt(1.0 / knee), np.sqrt(1.0 / freqs)) / np.sqrt(1.0 / knee)
    k /= np.sum(k[k > 0])
    k /= n_possible / len(harmonics)
    return k


def swipe_tables(sample_rate=SAMPLING_RATE, min_freq=65, max_freq=1047, cres=10):
    """(log-spaced frequency grid, kernel matrix [candidates][grid]) — friture/pitch_tracker.py:334-355."""
    n = int(np.log2(sample_rate / (2 * min_freq)) * (1200 / cres))
    freqs = np.logspace(np.log2(min_freq), np.log2(sample_rate // 2), num=n, base=2)
    n_cand = int(np.searchsorted(freqs, max_freq))
    kernels = np.zeros((n_cand, n))
    for i in range(n_cand):
        kernels[i] = swipe_kernel(freqs[i], freqs)
    return freqs, kernels


def pitch_strengths(frame, window, freqs, kernels, sample_rate=SAMPLING_RATE):
    """|rfft(frame*window)| -> np.interp onto the log grid -> RMS normalise -> kernels @ spectrum
    — friture/pitch_tracker.py:370-383."""
    n_fft = len(frame)
    spectrum = np.abs(np.fft.rfft(frame * window))
    lin = np.arange(len(spectrum), dtype="float64") * (float(sample_rate) / float(n_fft))
    spec_log = np.interp(freqs, lin, spectrum)
    spec_log = spec_log / np.sqrt(np.mean(spec_log ** 2))
    return np.matmul(kernels, spec_log)


def pitch_candidate(frame, window, freqs, kernels, sample_rate=SAMPLING_RATE):
    """(f0 before gating, confidence, dBFS) of one frame — friture/pitch_tracker.py:370-420."""
    s = pitch_strengths(frame, window, freqs, kernels, sample_rate)
    i = int(np.argmax(s))
    shift = parabolic_vertex(s[i - 1], s[i], s[i + 1])[0] if 0 < i < len(s) - 1 else 0
    f0 = np.interp(i + shift, np.arange(len(freqs)), freqs)
    rms = np.sqrt(np.mean(np.asarray(frame, np.float64) ** 2))
    return float(f0), float(s[i] / 2.56), float(20 * np.log10(rms + np.finfo(np.float64).eps))


class PitchGate:
    """The voiced/unvoiced decision and its one word of state — friture/pitch_tracker.py:405-428:
    unvoiced (nan) when the frame is quieter than min_db, the confidence is under `conf`, or the
    estimate jumped more than p_delta semitones from the previous *voiced* estimate."""

    def __init__(self, min_db=-50.0, conf=0.5, p_delta=2):
        self.min_db, self.conf, self.p_delta = min_db, conf, p_delta
        self.prev = None

    def step(self, f0, confidence, dbfs):
        jump = 12 * np.abs(np.log2(f0 / self.prev)) if self.prev is not None else 0
        if dbfs < self.min_db or confidence < self.conf or jump > self.p_delta:
            self.prev = None
            return np.nan
        self.prev = f0
        return f0


def pitch_track(x, n_fft, hop, freqs, kernels, sample_rate=SAMPLING_RATE, min_db=-50.0, conf=0.5, p_delta=2):
    """All complete frames of x at the given hop through candidate + gate
    — friture/pitch_tracker.py:313-332 (update/new_frames).  Returns (f0 with nan = unvoiced,
    raw f0, confidence, dBFS), one entry per frame."""
    window = hann_symmetric(n_fft)
    gate = PitchGate(min_db, conf, p_delta)
    frames = 0 if len(x) < n_fft else (len(x) - n_fft) // hop + 1
    out = np.zeros((4, frames))
    for g in range(frames):
        f0, c, db = pitch_candidate(np.asarray(x[g * hop:g * hop + n_fft], np.float64), window, freqs, kernels, sample_rate)
        out[:, g] = gate.step(f0, c, db), f0, c, db
    return out
