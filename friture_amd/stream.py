"""Streaming STFT front-end with a device-resident sample window.

The reference's widgets re-read overlapping windows from a host ring buffer for every frame
(friture/ringbuffer.py:87-99, friture/spectrogram.py:149-159).  Here the unconsumed tail of the
stream lives in HBM: `push(chunk)` uploads only the new samples, appends them to the tail on the
device, runs the batched STFT kernel over every frame that became realizable, and keeps the samples
the next frame still needs.  Results are identical to one batch transform of the whole stream
(same kernel, same frame alignment); the frames of a push stay in HBM.
"""
from __future__ import annotations

import numpy as np

from .stft import StftEngine


class StftStream:
    """The unconsumed samples live in a mirror ring in HBM (the layout of friture/ringbuffer.py:39-63: every sample is
    stored at p and p + L, so any window of up to L samples is one contiguous slice): a push writes the new samples twice
    and transforms, in place, every frame that became realizable — nothing is shifted or copied afterwards."""

    def __init__(self, fft_size: int, hop: int, n_channels: int = 1, max_chunk: int = 1 << 16):
        import torch
        self._torch = torch
        self.fft_size, self.hop, self.n_channels = fft_size, hop, n_channels
        self.engine = StftEngine(fft_size, hop, n_channels, 32)
        self._dev = torch.device("cuda", torch.cuda.current_device())
        self._len = fft_size + max_chunk + hop                 # samples the ring retains
        self._len += self._len & 1                             # even: windows keep the parity of their stream index
        self._buf = torch.zeros((n_channels, 2 * self._len), dtype=torch.float32, device=self._dev)
        self._offset = 0                                       # samples pushed so far
        self._consumed = 0                                     # stream index of the first sample the next frame needs
        self.frames_emitted = 0

    def set_epilogue(self, *args, **kw):
        self.engine.set_epilogue(*args, **kw)

    def _grow(self, need):
        torch = self._torch
        old_len, new_len = self._len, int(1.5 * need)
        new_len += new_len & 1
        grown = torch.zeros((self.n_channels, 2 * new_len), dtype=torch.float32, device=self._dev)
        keep = self._offset - self._consumed                   # the unconsumed tail, re-laid at its new positions
        if keep:
            tail = self._window(self._consumed, keep).clone()
            self._len, self._buf = new_len, grown
            self._write(self._consumed, tail)
        else:
            self._len, self._buf = new_len, grown

    def _window(self, start, length):
        p = start % self._len
        return self._buf[:, p:p + length]

    def _write(self, start, data):
        n, L = data.shape[1], self._len
        p = start % L
        straight = min(n, L - p)
        self._buf[:, p:p + straight].copy_(data[:, :straight], non_blocking=True)
        self._buf[:, p + L:p + L + straight].copy_(data[:, :straight], non_blocking=True)
        if n > straight:
            self._buf[:, :n - straight].copy_(data[:, straight:], non_blocking=True)
            self._buf[:, L:L + n - straight].copy_(data[:, straight:], non_blocking=True)

    def push(self, chunk, kind: int = 0):
        """chunk: [C, n] float32 (numpy, uploaded; or a CUDA tensor).  Returns the new frames
        [C, F, N/2+1] as a CUDA tensor (F may be 0)."""
        torch = self._torch
        if not torch.is_tensor(chunk):
            chunk = torch.from_numpy(np.ascontiguousarray(chunk, np.float32))
        if chunk.dim() == 1:
            chunk = chunk[None, :]
        n = chunk.shape[1]
        if (self._offset - self._consumed) + n > self._len:
            self._grow((self._offset - self._consumed) + n)
        self._write(self._offset, chunk.to(self._dev, non_blocking=True) if not chunk.is_cuda else chunk)
        self._offset += n
        avail = self._offset - self._consumed
        frames = self.engine.frames_for(avail)
        bins = self.fft_size // 2 + 1
        out_dtype = torch.int32 if kind == 3 else torch.float32
        out = torch.empty((self.n_channels, frames, bins), dtype=out_dtype, device=self._dev)
        if frames:
            span = self.fft_size + (frames - 1) * self.hop
            self._run_strided(kind, self._window(self._consumed, span), out)
            self._consumed += frames * self.hop
            self.frames_emitted += frames
        return out

    def _run_strided(self, kind, view, out):
        import ctypes

        from . import _lib
        torch = self._torch
        eng = self.engine
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(eng._lib.frt_stft_set_stream(eng._h, ctypes.c_void_p(stream)))
        nf = ctypes.c_int64(0)
        _lib.check(eng._lib.frt_stft_run(eng._h, kind, ctypes.c_void_p(view.data_ptr()), view.shape[1], view.stride(0),
                                         ctypes.c_void_p(out.data_ptr()), ctypes.byref(nf)))
