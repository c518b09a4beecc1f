"""friture/filter.py on the GPU: the exact IIR octave bank with decimation (legacy path).

`octave_filter_bank_decimation(blow, alow, forward, feedback, x, zis) -> (y, dec, zfs)` and
`octave_filter_bank_decimation_filtic(...)` keep the reference's functional signatures
(friture/filter.py:86-133); `octave_frequencies` and `NOCTAVE` are re-exported as the reference does.
Every call runs the nine octave stages of kernel K2 on a cached bank handle: states go in through
frt_octbank_set_state, band signals come back packed and are split into the reference's list.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .constants import NOCTAVE  # noqa: F401  (re-export, friture/filter.py:7)

_DP = ctypes.POINTER(ctypes.c_double)
_banks: dict = {}


def octave_frequencies(total_bands_count, bands_per_octave):
    """Centre and edge frequencies fi = 1000 * 2^(i / bpo), i symmetric around 1 kHz."""
    half = total_bands_count // 2
    idx = np.arange(-half, half) if total_bands_count % 2 == 0 else np.arange(-half, half + 1)
    b = 1. / bands_per_octave
    fi = 1000. * 2 ** (idx * b)
    return fi, fi * np.sqrt(2 ** (-b)), fi * np.sqrt(2 ** b)


class IirBank:
    """One device-resident exact IIR bank (any number of channels)."""

    def __init__(self, blow, alow, forward, feedback, n_channels=1):
        lib = _lib.init()
        self._lib = lib
        self.bpo = len(forward)
        self.nbands = NOCTAVE * self.bpo
        self.n_channels = n_channels
        boct = np.ascontiguousarray(np.asarray(forward, np.float64))
        aoct = np.ascontiguousarray(np.asarray(feedback, np.float64))
        if boct.shape != (self.bpo, 5) or aoct.shape != (self.bpo, 5):
            raise ValueError("band-pass filters must be 4th order (5 + 5 coefficients)")
        blow = np.ascontiguousarray(blow, np.float64)
        alow = np.ascontiguousarray(alow, np.float64)
        if blow.shape != (13,) or alow.shape != (13,):
            raise ValueError("decimation filter must be 12th order (13 + 13 coefficients)")
        self._h = ctypes.c_void_p()
        _lib.check(lib.frt_octbank_create(ctypes.byref(self._h), self.bpo, n_channels, 0, boct.ctypes.data_as(_DP),
                                          aoct.ctypes.data_as(_DP), blow.ctypes.data_as(_DP), alow.ctypes.data_as(_DP),
                                          None, None))
        self.state_length = lib.frt_octbank_state_length(self._h)

    def __del__(self):
        try:
            if self._h.value:
                self._lib.frt_octbank_destroy(self._h)
        except Exception:
            pass

    def reset(self):
        _lib.check(self._lib.frt_octbank_reset(self._h))

    def set_chunk(self, chunk0: int):
        _lib.check(self._lib.frt_octbank_set_chunk(self._h, int(chunk0)))

    def set_state(self, z):
        z = np.ascontiguousarray(z, np.float64).reshape(self.n_channels, self.state_length)
        _lib.check(self._lib.frt_octbank_set_state(self._h, z.ctypes.data_as(_DP)))

    def get_state(self):
        z = np.empty((self.n_channels, self.state_length), np.float64)
        _lib.check(self._lib.frt_octbank_get_state(self._h, z.ctypes.data_as(_DP)))
        return z

    def band_lengths(self, n):
        lens = [n]
        for _ in range(NOCTAVE - 1):
            lens.append((lens[-1] + 1) // 2)
        return [lens[NOCTAVE - 1 - k // self.bpo] for k in range(self.nbands)]

    def filter(self, x):
        """x: [C, n] float64 -> (list over channels of list over bands of arrays, dec list)."""
        x = np.ascontiguousarray(x, np.float64)
        if x.ndim == 1:
            x = x[None, :]
        n = x.shape[1]
        if n == 0:
            raise Exception("Filter input is too small")
        plen = self._lib.frt_octbank_packed_length(self._h, n)
        packed = np.empty((self.n_channels, plen), np.float64)
        dec = (ctypes.c_int * self.nbands)()
        _lib.check(self._lib.frt_octbank_filter(self._h, x.ctypes.data, n, packed.ctypes.data, dec))
        offs = np.concatenate([[0], np.cumsum(self.band_lengths(n))])
        bands = [[packed[c, offs[k]:offs[k + 1]] for k in range(self.nbands)] for c in range(self.n_channels)]
        return bands, list(dec)

    def energies(self, x, block, alphas, weight_db=None, as_db=False, out=None):
        """Band energies per block (frt_octbank_energies).  x: float32 [C, n] numpy or torch CUDA tensor."""
        al = np.ascontiguousarray(alphas, np.float64)
        wp = None
        if weight_db is not None:
            w = np.ascontiguousarray(weight_db, np.float64)
            wp = w.ctypes.data_as(_DP)
        if type(x).__module__.startswith("torch"):
            import torch
            assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
            n = x.shape[1]
            if out is None:
                out = torch.empty((self.n_channels, n // block, self.nbands), dtype=torch.float32, device=x.device)
            _lib.check(self._lib.frt_octbank_set_stream(self._h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
            _lib.check(self._lib.frt_octbank_energies(self._h, ctypes.c_void_p(x.data_ptr()), n, block, al.ctypes.data_as(_DP),
                                                      wp, int(as_db), ctypes.c_void_p(out.data_ptr())))
            return out
        x = np.ascontiguousarray(x, np.float32)
        if x.ndim == 1:
            x = x[None, :]
        n = x.shape[1]
        if out is None:
            out = np.empty((self.n_channels, n // block, self.nbands), np.float32)
        _lib.check(self._lib.frt_octbank_energies(self._h, x.ctypes.data, n, block, al.ctypes.data_as(_DP), wp, int(as_db),
                                                  out.ctypes.data))
        return out


class FirBank(IirBank):
    """One device-resident FFT overlap-add bank (the production bank of Octave_Filters.filter,
    friture/octavefilters.py:49-58 -> friture/filter.py:136-247) for any number of channels and any batch
    length: `filter(x)` / `energies(x, ...)` return what the reference returns when x is fed to it in blocks
    of at most 1024 samples (512-tap minimum-phase FIRs of every IIR, 511-sample tails carried across calls).
    Up to 1024 samples a call is the reference's own overlap-add block; longer inputs run every octave stage
    as ONE launch over all blocks (ola_batch_kernel)."""

    def __init__(self, bands_per_octave, n_channels=1, tables=None):
        from . import filter_design
        lib = _lib.init()
        self._lib = lib
        t = tables if tables is not None else filter_design.load_tables()
        if "boct_%d" % bands_per_octave not in t:
            raise Exception("Unknown bandsperoctave: %d" % (bands_per_octave))
        self.bpo = bands_per_octave
        self.nbands = NOCTAVE * self.bpo
        self.n_channels = n_channels
        boct = np.ascontiguousarray(np.asarray(t["boct_%d" % self.bpo], np.float64))
        aoct = np.ascontiguousarray(np.asarray(t["aoct_%d" % self.bpo], np.float64))
        bdec = np.ascontiguousarray(t["bdec"], np.float64)
        adec = np.ascontiguousarray(t["adec"], np.float64)
        fir = np.ascontiguousarray(np.asarray(t["boct_fir_%d" % self.bpo], np.float64))
        fird = np.ascontiguousarray(t["bdec_fir"], np.float64)
        self._h = ctypes.c_void_p()
        _lib.check(lib.frt_octbank_create(ctypes.byref(self._h), self.bpo, n_channels, 1, boct.ctypes.data_as(_DP),
                                          aoct.ctypes.data_as(_DP), bdec.ctypes.data_as(_DP), adec.ctypes.data_as(_DP),
                                          fir.ctypes.data_as(_DP), fird.ctypes.data_as(_DP)))
        self.state_length = 0

    def set_chunk(self, chunk0: int):
        raise NotImplementedError("the FIR bank has no recurrence to chunk")

    def set_state(self, z):
        raise NotImplementedError("the FIR bank's state is its pending tails: reset() zeroes them")

    get_state = set_state


def _bank_for(blow, alow, forward, feedback):
    key = (np.asarray(blow, np.float64).tobytes(), np.asarray(alow, np.float64).tobytes(),
           np.asarray(forward, np.float64).tobytes(), np.asarray(feedback, np.float64).tobytes())
    bank = _banks.get(key)
    if bank is None:
        bank = _banks[key] = IirBank(blow, alow, forward, feedback, 1)
    return bank


def octave_filter_bank_decimation(blow, alow, forward, feedback, x, zis):
    """Filter x with the bank; zis / zfs are lists of state vectors in processing order: for each
    octave the band filters from the highest band down, then the decimation filter."""
    bank = _bank_for(blow, alow, forward, feedback)
    bank.set_state(np.concatenate([np.asarray(z, np.float64) for z in zis]))
    bands, dec = bank.filter(x)
    flat = bank.get_state()[0]
    zfs, pos = [], 0
    for z in zis:
        zfs.append(flat[pos:pos + len(z)].copy())
        pos += len(z)
    return bands[0], dec, zfs


def octave_filter_bank_decimation_filtic(blow, alow, forward, feedback):
    """Zero initial conditions, in the order octave_filter_bank_decimation consumes them."""
    zfs = []
    for _ in range(NOCTAVE):
        for i in range(len(forward))[::-1]:
            zfs.append(np.zeros(max(len(forward[i]), len(feedback[i])) - 1))
        zfs.append(np.zeros(max(len(blow), len(alow)) - 1))
    return zfs
