"""A chain of processing stages with the block protocol of friture/signal/transform_pipeline.py:23-34:
every stage exposes push(columns) -> columns, and the chain itself is such a stage.  The spectrogram
widget builds it from the frequency resampler, the online time resampler and the colour transform
(friture/spectrogram.py:62-68) and reaches into `.blocks` to reconfigure them.

When the blocks are exactly those three (this package's classes), push() runs them as ONE device call
(frt_screen_columns: np.interp -> lerp against the carried column -> clip + LUT, the blocks' own operations in their own
order, pixel for pixel what the three pushes give) instead of three host-staged ones; the blocks keep their state — the
time resampler's indices and carried column are advanced exactly as its own push() advances them.  Any other chain falls
back to the reference's reduce()."""
from functools import reduce

import numpy as np


class Transform_Pipeline:
    def __init__(self, blocks):
        self.blocks = blocks

    def _fusable(self):
        from .color_tranform import Color_Transform
        from .frequency_resampler import Frequency_Resampler
        from .online_linear_2D_resampler import Online_Linear_2D_resampler
        b = self.blocks
        return (len(b) == 3 and type(b[0]) is Frequency_Resampler and type(b[1]) is Online_Linear_2D_resampler
                and type(b[2]) is Color_Transform)

    def push(self, data):
        if not self._fusable():
            return reduce(lambda columns, stage: stage.push(columns), self.blocks, data)
        fr, tr, ct = self.blocks
        data = np.asarray(data, np.float64)
        if data.ndim != 2 or data.shape[1] == 0:
            return reduce(lambda columns, stage: stage.push(columns), self.blocks, data)
        freq = np.ascontiguousarray(fr.freq, np.float64)
        targets = np.ascontiguousarray(fr.xscaled, np.float64)
        if data.shape[0] != freq.size:
            raise ValueError("fp and xp are not of the same length.")          # numpy.interp's complaint
        height, n_cols = targets.size, data.shape[1]
        tr.set_height(height)                                  # Fourier-resamples the carried column on a resize
        total, src, a = tr.advance(n_cols)                     # the scalar index recurrence of Online_Linear_2D_resampler.push
        old_in = np.ascontiguousarray(tr.old_data, np.float64)
        norm = np.ascontiguousarray(data.T)                    # frame-major: a column of the block is a row here
        lut = np.ascontiguousarray(ct.colors, np.uint32)
        n_out = len(src)
        pix = np.empty((height, max(n_out, 1)), np.uint32)
        old_out = np.empty(height)
        from .. import _lib
        _lib.check(fr._lib.frt_screen_columns(norm.ctypes.data, freq.size, n_cols, freq.ctypes.data, targets.ctypes.data, height,
                                              old_in.ctypes.data, src.ctypes.data if n_out else None, a.ctypes.data if n_out else None,
                                              n_out, lut.ctypes.data, pix.ctypes.data if n_out else None, old_out.ctypes.data))
        tr.old_data = old_out
        out = np.full((height, max(total, 0)), lut[0], np.uint32)              # columns the resampler left at zero map to lut[0]
        if n_out:
            out[:, :n_out] = pix[:, :n_out]
        return out
