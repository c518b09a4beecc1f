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
    def __init__(self, fft_size: int, hop: int, n_channels: int = 1, max_chunk: int = 1 << 16):
        import torch
        self._torch = torch
        self.fft_size, self.hop, self.n_channels = fft_size, hop, n_channels
        self.engine = StftEngine(fft_size, hop, n_channels, 32)
        self._dev = torch.device("cuda", torch.cuda.current_device())
        self._cap = fft_size + max_chunk + hop
        self._buf = torch.zeros((n_channels, self._cap), dtype=torch.float32, device=self._dev)
        self._fill = 0                       # valid samples at the head of the buffer
        self.frames_emitted = 0

    def set_epilogue(self, *args, **kw):
        self.engine.set_epilogue(*args, **kw)

    def push(self, chunk, kind: int = 0):
        """chunk: [C, n] float32 (numpy, uploaded; or a CUDA tensor).  Returns the new frames
        [C, F, N/2+1] as a CUDA tensor (F may be 0)."""
        torch = self._torch
        if not torch.is_tensor(chunk):
            chunk = torch.from_numpy(np.ascontiguousarray(chunk, np.float32))
        if chunk.dim() == 1:
            chunk = chunk[None, :]
        n = chunk.shape[1]
        if self._fill + n > self._cap:       # grow the device window like the host ring does (x1.5)
            cap = int(1.5 * (self._fill + n))
            grown = torch.zeros((self.n_channels, cap), dtype=torch.float32, device=self._dev)
            grown[:, :self._fill] = self._buf[:, :self._fill]
            self._buf, self._cap = grown, cap
        self._buf[:, self._fill:self._fill + n].copy_(chunk, non_blocking=True)
        self._fill += n
        frames = self.engine.frames_for(self._fill)
        bins = self.fft_size // 2 + 1
        out_dtype = torch.int32 if kind == 3 else torch.float32
        out = torch.empty((self.n_channels, frames, bins), dtype=out_dtype, device=self._dev)
        if frames:
            # the engine wants a contiguous [C, T] view: run on the filled prefix via the row stride
            view = self._buf[:, :self._fill]
            self._run_strided(kind, view, out)
            consumed = frames * self.hop
            keep = self._fill - consumed
            self._buf[:, :keep] = self._buf[:, consumed:self._fill].clone()
            self._fill = keep
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
