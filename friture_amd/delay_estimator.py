"""The delay estimator's processing chain (friture/delay_estimator.py:87-176) without its Qt shell.

`DelayEstimator.handle_new_data(floatdata)` does what the widget's slot does for a two-channel
chunk: two chained IIR decimations per channel with carried state (kernel K2 via
friture_amd.signal.decimate), private ring buffers of the 12 kHz signals, 50 %-overlapped windows of
2 * delayrange * 12000 samples, GCC-PHAT per window (kernel K5), exponential smoothing of the
correlation, peak pick and the delay / polarity / confidence read-out (frt_gcc_readout).
"""
from __future__ import annotations

import numpy as np

from . import filter_design
from .constants import SAMPLING_RATE
from .ringbuffer import RingBuffer
from .signal.correlation import GccPhat, generalized_cross_correlation
from .signal.decimate import decimate_multiple, decimate_multiple_filtic

DEFAULT_DELAYRANGE = 1      # default delay range is 1 second (delay_estimator.py:31)


class DelayEstimator:
    def __init__(self, delayrange_s: float = DEFAULT_DELAYRANGE):
        t = filter_design.load_tables()
        self.Ndec = 2
        self.subsampled_sampling_rate = SAMPLING_RATE / 2 ** self.Ndec
        self.bdec, self.adec = np.array(t["bdec"]), np.array(t["adec"])
        self.zfs0 = decimate_multiple_filtic(self.Ndec, self.bdec, self.adec)
        self.zfs1 = decimate_multiple_filtic(self.Ndec, self.bdec, self.adec)
        self.ringbuffer0, self.ringbuffer1 = RingBuffer(), RingBuffer()
        self.delayrange_s = delayrange_s
        self.old_Xcorr = None
        self.old_index = 0
        self.two_channels = False
        self.delay_ms = 0.
        self.distance_m = 0.
        self.correlation = 0.
        self.Xcorr_extremum = 0.
        self._gcc = None

    def set_delayrange(self, delay_s):
        self.delayrange_s = delay_s

    def handle_new_data(self, floatdata):
        if floatdata.shape[0] == 1:
            self.two_channels = False
            return
        self.two_channels = True
        x0_dec, self.zfs0 = decimate_multiple(self.Ndec, self.bdec, self.adec, floatdata[0, :], self.zfs0)
        x1_dec, self.zfs1 = decimate_multiple(self.Ndec, self.bdec, self.adec, floatdata[1, :], self.zfs1)
        self.ringbuffer0.push(x0_dec.reshape(1, -1), 0)
        self.ringbuffer1.push(x1_dec.reshape(1, -1), 0)

        index = self.ringbuffer0.offset
        available = index - self.old_index
        if available < 0:
            available = 0
            self.old_index = index
        time = 2 * self.delayrange_s
        length = int(time * self.subsampled_sampling_rate)
        needed = int(0.5 * length)
        for _ in range(int(available / needed)):
            self.old_index += needed
            d0 = self.ringbuffer0.data_indexed(self.old_index, length).reshape(-1)     # views into the rings
            d1 = self.ringbuffer1.data_indexed(self.old_index, length).reshape(-1)
            if np.std(d0) > 0. and np.std(d1) > 0.:
                Xcorr = generalized_cross_correlation(d0, d1)                            # de-means the views in place
                if self._gcc is None or self._gcc.length != length:
                    self._gcc = GccPhat(length, 1)
                old = self.old_Xcorr if self.old_Xcorr is not None and self.old_Xcorr.shape == Xcorr.shape else None
                smoothed, ro = self._gcc.readout(Xcorr, old, self.subsampled_sampling_rate, self.delayrange_s, 0.3)
                self.old_Xcorr = smoothed[0]
                self.Xcorr_extremum = ro[0].extremum
                self.delay_ms = ro[0].delay_ms
                self.distance_m = ro[0].distance_m
                self.correlation = ro[0].correlation_pct
            else:
                self.delay_ms = 0.
                self.Xcorr_extremum = 0.
                self.distance_m = 0.
                self.correlation = 0


class DelayEstimatorStream(DelayEstimator):
    """The same chain with its signals resident in HBM: the two-stage decimation runs on a two-channel bank object whose
    filter states stay on the device, the 12 kHz rings are DeviceRingBuffers (same layout and growth as the host ring),
    windows go to GCC-PHAT and the read-out as device slices, and the in-place mean removal the reference applies to its
    ring views (correlation.py:27-28) is applied to the same slices.  Per chunk only the new samples go up; per window
    two scalars (the std test of delay_estimator.py:127-129) and the read-out come down."""

    def __init__(self, delayrange_s: float = DEFAULT_DELAYRANGE):
        super().__init__(delayrange_s)
        import ctypes

        import torch

        from . import _lib
        from .ringbuffer import DeviceRingBuffer
        self._torch, self._ct, self._libmod = torch, ctypes, _lib
        self._lib = _lib.init()
        self.ringbuffer0, self.ringbuffer1 = DeviceRingBuffer(), DeviceRingBuffer()
        self._dec = ctypes.c_void_p()
        DP = ctypes.POINTER(ctypes.c_double)
        b, a = np.ascontiguousarray(self.bdec, np.float64), np.ascontiguousarray(self.adec, np.float64)
        _lib.check(self._lib.frt_octbank_create(ctypes.byref(self._dec), 0, 2, 0, None, None, b.ctypes.data_as(DP), a.ctypes.data_as(DP),
                                                None, None))
        self._dev = torch.device("cuda", torch.cuda.current_device())

    def __del__(self):
        try:
            if self._dec.value:
                self._lib.frt_octbank_destroy(self._dec)
        except Exception:
            pass

    def handle_new_data(self, floatdata):
        torch, ct, check = self._torch, self._ct, self._libmod.check
        if floatdata.shape[0] == 1:
            self.two_channels = False
            return
        self.two_channels = True
        x = torch.from_numpy(np.ascontiguousarray(floatdata[:2], np.float64)).to(self._dev)
        n = x.shape[1]
        n_out = ct.c_int(0)
        dec = torch.empty((2, (n + 3) // 4 + 1), dtype=torch.float64, device=self._dev)
        stream = ct.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._lib.frt_octbank_set_stream(self._dec, stream))
        packed = torch.empty((2 * ((n + 3) // 4 + 1),), dtype=torch.float64, device=self._dev)
        check(self._lib.frt_decimate_multiple(self._dec, self.Ndec, ct.c_void_p(x.data_ptr()), n, ct.c_void_p(packed.data_ptr()),
                                              ct.byref(n_out)))
        m = n_out.value
        dec = packed[:2 * m].view(2, m)
        self.ringbuffer0.push(dec[0:1], 0)
        self.ringbuffer1.push(dec[1:2], 0)

        index = self.ringbuffer0.offset
        available = index - self.old_index
        if available < 0:
            available = 0
            self.old_index = index
        time = 2 * self.delayrange_s
        length = int(time * self.subsampled_sampling_rate)
        needed = int(0.5 * length)
        for _ in range(int(available / needed)):
            self.old_index += needed
            d0 = self.ringbuffer0.data_indexed(self.old_index, length)          # [1, length] device views into the rings
            d1 = self.ringbuffer1.data_indexed(self.old_index, length)
            stds = torch.stack([d0.std(unbiased=False), d1.std(unbiased=False)]).cpu()
            if stds[0] > 0. and stds[1] > 0.:
                if self._gcc is None or self._gcc.length != length:
                    self._gcc = GccPhat(length, 1)
                Xcorr, _ = self._gcc.correlate(d0.contiguous(), d1.contiguous())
                d0 -= self._gcc.means[0, 0]                                     # the reference de-means its views in place
                d1 -= self._gcc.means[0, 1]
                old = self.old_Xcorr if self.old_Xcorr is not None and self.old_Xcorr.shape == Xcorr.shape else None
                smoothed, ro = self._gcc.readout(Xcorr, old, self.subsampled_sampling_rate, self.delayrange_s, 0.3)
                self.old_Xcorr = smoothed
                self.Xcorr_extremum = ro[0].extremum
                self.delay_ms = ro[0].delay_ms
                self.distance_m = ro[0].distance_m
                self.correlation = ro[0].correlation_pct
            else:
                self.delay_ms = 0.
                self.Xcorr_extremum = 0.
                self.distance_m = 0.
                self.correlation = 0
