"""Drop-in for the reference's `friture.octavefilters.Octave_Filters` (friture/octavefilters.py:37-158).

Attribute surface kept: `filter(floatdata) -> (list of 9*bpo arrays, list of
decimation factors)`, `get_decs()`, `setbandsperoctave(b)`, attributes `.bdec .adec .boct .aoct
.fi .flow .fhigh .f_nominal .A .B .C .nbands .bandsperoctave .FIR_LENGTH`, module constant
`NOCTAVE`.  `filter` runs the FFT overlap-add bank of kernel K3 on the GPU (mode 1 of
frt_octbank_*), with the pending tails carried on the device between calls; the exact IIR bank the
FIRs were derived from lives in friture_amd.filter.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib, filter_design, renard, tables
from .constants import FIR_LENGTH, NOCTAVE  # noqa: F401  (NOCTAVE re-exported as in the reference)
from .filter import octave_frequencies

_DP = ctypes.POINTER(ctypes.c_double)


def nominal_labels(fi, bandsperoctave):
    """Band labels: preferred numbers around the 1 kHz reference, 'k' suffix from 1 kHz up."""
    if bandsperoctave == 1:
        return ["%.1fk" % (f / 1000) if f >= 10000 else "%.2fk" % (f / 1000) if f >= 1000 else "%d" % f for f in fi]
    if bandsperoctave not in renard.SERIES:
        raise Exception("Unknown bandsperoctave: %d" % (bandsperoctave))
    series = renard.SERIES[bandsperoctave]
    ref = int(np.where(np.asarray(fi) == 1000.)[0][0])
    n_up, n_down = len(fi) - ref, ref
    up, decade = [], 0
    while len(up) < n_up:                       # 1.00k .. 9.xxk, then 10.0k .., one decade per turn
        up += ["{0:.{width}f}k".format(10 ** decade * v, width=2 - decade) for v in series]
        decade += 1
    down, decade = [], 0
    while len(down) < n_down:                   # 100 .. 9xx, then 10 .. 99, ...
        down = ["%d" % (10 ** (2 - decade) * v) for v in series] + down
        decade += 1
    return (down + up[:n_up])[-len(fi):] if n_down else up[:n_up]


class Octave_Filters:

    FIR_LENGTH = FIR_LENGTH

    def __init__(self, bandsperoctave):
        self._lib = _lib.init()
        self._tables = filter_design.load_tables()
        self._h = ctypes.c_void_p()
        self.bdec = np.array(self._tables["bdec"])
        self.adec = np.array(self._tables["adec"])
        self.setbandsperoctave(bandsperoctave)

    # ---- the hot call -------------------------------------------------------------------------
    def filter(self, floatdata):
        x = np.ascontiguousarray(floatdata, np.float64)
        n = x.shape[0]
        if n == 0:
            raise Exception("Filter input is too small")
        plen = self._lib.frt_octbank_packed_length(self._h, n)
        packed = np.empty(plen, np.float64)
        dec = (ctypes.c_int * self.nbands)()
        _lib.check(self._lib.frt_octbank_filter(self._h, x.ctypes.data, n, packed.ctypes.data, dec))
        lens = [n]
        for _ in range(NOCTAVE - 1):
            lens.append((lens[-1] + 1) // 2)
        y, pos = [], 0
        for k in range(self.nbands):
            m = lens[NOCTAVE - 1 - k // self.bandsperoctave]
            y.append(packed[pos:pos + m])
            pos += m
        # the band signals lie back to back in `packed`, the bands of an octave equally long: kept for callers that hand whole
        # octaves on (OctaveSpectrum: one smoothing call per chunk) — valid until the next call
        self._packed, self._packed_lens = packed, lens
        return y, list(dec)

    def get_decs(self):
        return [2 ** j for j in range(0, NOCTAVE)[::-1] for _ in range(0, self.bandsperoctave)]

    def reset(self):
        """Zero the pending overlap tails (what _init_fir_and_states does on a band-count change)."""
        _lib.check(self._lib.frt_octbank_reset(self._h))

    # ---- configuration --------------------------------------------------------------------------
    def setbandsperoctave(self, bandsperoctave):
        if "boct_%d" % bandsperoctave not in self._tables:
            raise Exception("Unknown bandsperoctave: %d" % (bandsperoctave))
        self.bandsperoctave = bandsperoctave
        self.nbands = NOCTAVE * bandsperoctave
        self.fi, self.flow, self.fhigh = octave_frequencies(self.nbands, bandsperoctave)
        self.boct = [np.array(f) for f in self._tables["boct_%d" % bandsperoctave]]
        self.aoct = [np.array(f) for f in self._tables["aoct_%d" % bandsperoctave]]
        self.A, self.B, self.C = tables.weighting_db(self.fi)
        self.f_nominal = nominal_labels(self.fi, bandsperoctave)
        self._boct_fir = list(self._tables["boct_fir_%d" % bandsperoctave])
        self._bdec_fir = self._tables["bdec_fir"]
        self._fft_sizes = [int(s) for s in self._tables["fft_sizes"]]
        self._release()
        boct = np.ascontiguousarray(np.asarray(self.boct, np.float64))
        aoct = np.ascontiguousarray(np.asarray(self.aoct, np.float64))
        fir = np.ascontiguousarray(np.asarray(self._boct_fir, np.float64))
        fird = np.ascontiguousarray(self._bdec_fir, np.float64)
        _lib.check(self._lib.frt_octbank_create(
            ctypes.byref(self._h), bandsperoctave, 1, 1, boct.ctypes.data_as(_DP), aoct.ctypes.data_as(_DP),
            self.bdec.ctypes.data_as(_DP), self.adec.ctypes.data_as(_DP), fir.ctypes.data_as(_DP), fird.ctypes.data_as(_DP)))

    def _release(self):
        if self._h.value:
            self._lib.frt_octbank_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass
