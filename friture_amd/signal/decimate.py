"""friture/signal/decimate.py:27-84 on the GPU: decimate, decimate_multiple, decimate_multiple_filtic.

Same functional interface as the reference (filter states are passed in and returned).  The chain
of `Ndec` low-pass + take-every-other stages runs as `Ndec` launches of the IIR stage kernel on a
decimator-only bank handle (frt_decimate_multiple); handles are cached per coefficient set.
"""
from __future__ import annotations

import ctypes

import numpy as np

from .. import _lib
from .lfilter import lfilter_float64_1D

_DP = ctypes.POINTER(ctypes.c_double)
_handles: dict = {}
_IN_PLACE_MAX = 30000         # samples per call that frt_decimate_multiple_state takes (256 KB in place)


def _chain_handle(bdec, adec, channels=1):
    key = (bdec.tobytes(), adec.tobytes(), channels)
    h = _handles.get(key)
    if h is None:
        lib = _lib.init()
        h = ctypes.c_void_p()
        _lib.check(lib.frt_octbank_create(ctypes.byref(h), 0, channels, 0, None, None, bdec.ctypes.data_as(_DP),
                                          adec.ctypes.data_as(_DP), None, None))
        _handles[key] = h
    return h


def decimate(bdec, adec, x, zi=None):
    if len(x) == 0:
        raise Exception("Filter input is too small")
    if zi is None:
        zi = np.zeros(max(len(bdec), len(adec)) - 1, dtype=np.float64)
    x_dec, zf = lfilter_float64_1D(bdec, adec, x, zi)
    return x_dec[::2], zf


def decimate_multiple(Ndec, bdec, adec, x, zis):
    """Decimate Ndec times by 2; zis is a list of Ndec state vectors (or None for zero state,
    in which case no state is returned)."""
    x = np.ascontiguousarray(x, np.float64)
    if x.size == 0 or Ndec == 0:          # nothing to filter: the reference's loop body never runs (decimate.py:56-71)
        return x, zis
    bdec = np.ascontiguousarray(bdec, np.float64)
    adec = np.ascontiguousarray(adec, np.float64)
    if len(bdec) != 13 or len(adec) != 13 or Ndec > 8:
        # not the bank's 12th-order decimator: chain single-filter calls
        out, zfs = x, []
        for i in range(Ndec):
            out, zf = decimate(bdec, adec, out, None if zis is None else zis[i])
            zfs.append(zf)
        return out, (None if zis is None else zfs)
    lib = _lib.init()
    h = _chain_handle(bdec, adec)
    n_out = ctypes.c_int(0)
    out = np.empty((len(x) + 1) // 2, np.float64)      # upper bound
    if len(x) <= _IN_PLACE_MAX:
        # samples and states in, decimated samples and states out: one call, one synchronisation
        zi = None if zis is None else np.ascontiguousarray(np.stack([np.asarray(z, np.float64) for z in zis[:Ndec]]))
        zf = None if zis is None else np.empty((Ndec, 12), np.float64)
        _lib.check(lib.frt_decimate_multiple_state(h, int(Ndec), x.ctypes.data, len(x), None if zi is None else zi.ctypes.data,
                                                   out.ctypes.data, ctypes.byref(n_out), None if zf is None else zf.ctypes.data))
        out = out[:n_out.value]
        return out, (None if zis is None else [zf[j].copy() for j in range(Ndec)])
    slen = lib.frt_octbank_state_length(h)
    state = np.zeros(slen, np.float64)
    if zis is not None:
        for j, z in zip(range(Ndec), zis):
            state[12 * j:12 * (j + 1)] = z
    _lib.check(lib.frt_octbank_set_state(h, state.ctypes.data_as(_DP)))
    _lib.check(lib.frt_decimate_multiple(h, int(Ndec), x.ctypes.data, len(x), out.ctypes.data, ctypes.byref(n_out)))
    out = out[:n_out.value].copy()
    if zis is None:
        return out, None
    _lib.check(lib.frt_octbank_get_state(h, state.ctypes.data_as(_DP)))
    return out, [state[12 * j:12 * (j + 1)].copy() for j in range(Ndec)]


def decimate_multiple_channels(Ndec, bdec, adec, X, zis):
    """decimate_multiple for the rows of X [channels, n] in ONE device call (frt_decimate_multiple_state on a handle with that
    many channels): `zis` is a list, one entry per channel, of Ndec state vectors.  Returns (outs [channels, n_out], new zis) — every
    row what its own decimate_multiple call returns, bit for bit (the channels are independent slots of the same launches)."""
    X = np.ascontiguousarray(X, np.float64)
    C, n = X.shape
    bdec = np.ascontiguousarray(bdec, np.float64)
    adec = np.ascontiguousarray(adec, np.float64)
    if n == 0 or Ndec == 0 or len(bdec) != 13 or len(adec) != 13 or Ndec > 8 or C * n > _IN_PLACE_MAX:
        res = [decimate_multiple(Ndec, bdec, adec, X[c], zis[c]) for c in range(C)]
        return np.stack([r[0] for r in res]), [r[1] for r in res]
    lib = _lib.init()
    h = _chain_handle(bdec, adec, C)
    n_out = ctypes.c_int(0)
    n_up = n
    for _ in range(Ndec):
        n_up = (n_up + 1) // 2
    out = np.empty((C, n_up), np.float64)
    zi = np.ascontiguousarray(np.stack([np.stack([np.asarray(z, np.float64) for z in zis[c][:Ndec]]) for c in range(C)]))
    zf = np.empty((C, Ndec, 12), np.float64)
    _lib.check(lib.frt_decimate_multiple_state(h, int(Ndec), X.ctypes.data, n, zi.ctypes.data, out.ctypes.data, ctypes.byref(n_out),
                                               zf.ctypes.data))
    assert n_out.value == n_up
    return out, [[zf[c, j].copy() for j in range(Ndec)] for c in range(C)]


def decimate_multiple_filtic(Ndec, bdec, adec):
    """Zero initial conditions for the subsampler."""
    return [np.zeros(max(len(bdec), len(adec)) - 1) for _ in range(Ndec)]
