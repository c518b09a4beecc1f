"""Mirror ring buffer with the interface of friture/ringbuffer.py:28-130.

A ring of `buffer_length` samples stored twice back to back, so that any window of up to
`buffer_length` samples ending anywhere is one contiguous slice (`data`, `data_older`,
`data_indexed` return views, valid until the next push); the ring grows by 1.5x whenever a longer
window is requested.  This is pure data movement on the host side of the boundary: the analysis
kernels receive the contiguous windows it hands out.
"""
from __future__ import annotations

import numpy as np

from .constants import SAMPLING_RATE


class RingBuffer:
    """`zeros(shape)` makes the storage: numpy on the host (the default, what the widgets share), or — DeviceRingBuffer —
    a torch CUDA tensor, in which case pushes, windows and the growth are device-to-device slices and the windows handed
    out are device memory (same indices, same mirror layout, same growth rule: a view mutated in place, as
    generalized_cross_correlation does to its arguments, changes the same copy of the same samples)."""

    def __init__(self, zeros=None):
        self._zeros = zeros or (lambda shape: np.zeros(shape))
        self.buffer_length = 10000
        self.buffer = self._zeros((1, 2 * self.buffer_length))
        self.offset = 0
        self.offset_time = 0

    def push(self, floatdata, input_time: float = 0.) -> None:
        channels, count = floatdata.shape
        if channels != self.buffer.shape[0]:
            self.buffer = self._zeros((channels, 2 * self.buffer_length))   # mono <-> stereo switch starts afresh
        self.grow_if_needed(count)
        size = self.buffer_length
        head = self.offset % size
        self.buffer[:, head:head + count] = floatdata
        straight = min(count, size - head)                               # mirrored copy, wrapped at 2*size
        self.buffer[:, head + size:head + size + straight] = floatdata[:, :straight]
        self.buffer[:, :count - straight] = floatdata[:, straight:]
        self.offset += count
        self.offset_time = input_time

    def data(self, length):
        self.grow_if_needed(length)
        stop = self.offset % self.buffer_length + self.buffer_length
        return self._window(stop - length, stop)

    def data_older(self, length, delay_samples):
        self.grow_if_needed(length + delay_samples)
        start = (self.offset - length - delay_samples) % self.buffer_length + self.buffer_length
        return self.buffer[:, start:start + length]

    def data_indexed(self, start, length):
        """The `length` samples that end at absolute stream index `start`."""
        self.grow_if_needed(length + self.offset - start)
        stop = start % self.buffer_length + self.buffer_length
        return self._window(stop - length, stop)

    def data_time(self, start: int) -> float:
        return self.offset_time + (start - self.offset) / SAMPLING_RATE

    def _window(self, start, stop):
        if start > 2 * self.buffer_length or start < 0:
            raise ArithmeticError("Start index is wrong %d %d" % (start, self.buffer_length))
        if stop > 2 * self.buffer_length:
            raise ArithmeticError("Stop index is larger than buffer size: %d > %d" % (stop, 2 * self.buffer_length))
        return self.buffer[:, start:stop]

    def grow_if_needed(self, length):
        if length <= self.buffer_length:
            return
        old, new = self.buffer_length, int(1.5 * length)
        grown = self._zeros((self.buffer.shape[0], 2 * new))
        shift = (self.offset % new - self.offset % old) % new            # keeps self.offset meaningful
        grown[:, shift:shift + old] = self.buffer[:, :old]
        straight = min(old, new - shift)
        grown[:, new + shift:new + shift + straight] = self.buffer[:, :straight]
        grown[:, :old - straight] = self.buffer[:, straight:old]
        self.buffer, self.buffer_length = grown, new


class DeviceRingBuffer(RingBuffer):
    """The same ring in HBM (float64 torch tensor on the current CUDA device)."""

    def __init__(self):
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())
        super().__init__(lambda shape: torch.zeros(shape, dtype=torch.float64, device=dev))
