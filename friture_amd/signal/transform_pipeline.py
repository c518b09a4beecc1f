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
        self._fused_for = None          # ids of the blocks the fused path was last validated for
        self._tables = None             # (freq, targets, lut) as float64 / uint32 arrays + their addresses, keyed by identity

    def _fusable(self):
        b = self.blocks
        key = tuple(map(id, b))
        if self._fused_for is not None and self._fused_for[0] == key:
            return self._fused_for[1]
        from .color_tranform import Color_Transform
        from .frequency_resampler import Frequency_Resampler
        from .online_linear_2D_resampler import Online_Linear_2D_resampler
        ok = (len(b) == 3 and type(b[0]) is Frequency_Resampler and type(b[1]) is Online_Linear_2D_resampler
              and type(b[2]) is Color_Transform)
        self._fused_for = (key, ok)
        return ok

    def _constant_tables(self, fr, ct):
        """The frequency table, the screen rows' frequencies and the LUT as contiguous float64 / uint32 arrays and their
        addresses.  When a source array already has that form it is passed as it is (nothing to go stale); when a converted
        copy had to be made, the copy is kept for its address and REFRESHED from the source on every push — `ct.colors[:] = ...`
        or any other in-place change of the widget's arrays reaches the device call like it reaches the per-block pushes
        (ADVICE r3)."""
        t = self._tables
        if t is None or t[0] is not fr.freq or t[1] is not fr.xscaled or t[2] is not ct.colors:
            freq = np.ascontiguousarray(fr.freq, np.float64)
            targets = np.ascontiguousarray(fr.xscaled, np.float64)
            lut = np.ascontiguousarray(ct.colors, np.uint32)
            t = self._tables = (fr.freq, fr.xscaled, ct.colors, freq, targets, lut, freq.ctypes.data, targets.ctypes.data, lut.ctypes.data)
        else:
            for src, dst in ((t[0], t[3]), (t[1], t[4]), (t[2], t[5])):
                if dst is not src:
                    np.copyto(dst, src, casting="unsafe")
        return t[3:]

    def push(self, data):
        if not self._fusable():
            return reduce(lambda columns, stage: stage.push(columns), self.blocks, data)
        data = np.asarray(data, np.float64)
        if data.ndim != 2 or data.shape[1] == 0:
            return reduce(lambda columns, stage: stage.push(columns), self.blocks, data)
        norm = data.T                                          # frame-major: a column of the block is a row here
        if not norm.flags.c_contiguous:
            norm = np.ascontiguousarray(norm)
        return self._push_frames(norm.ctypes.data, data.shape[0], data.shape[1])

    def push_frames_device(self, norm_ptr, n_bins, n_cols):
        """push() for a block that is already on the device, frame-major [n_cols][n_bins] float64 (what the STFT kernel writes):
        the widget's chunk handler then costs ONE wait — transform enqueued, columns computed behind it (frt_screen_columns
        launches on the null stream, which is ordered behind the blocking stream the transform ran on).  Only for the fusable
        chain; the caller checks `fusable()`."""
        return self._push_frames(norm_ptr, n_bins, n_cols)

    def fusable(self):
        return self._fusable()

    def _push_frames(self, norm_ptr, n_bins, n_cols):
        fr, tr, ct = self.blocks
        freq, targets, lut, p_freq, p_targets, p_lut = self._constant_tables(fr, ct)
        if n_bins != freq.size:
            raise ValueError("fp and xp are not of the same length.")          # numpy.interp's complaint
        height = targets.size
        tr.set_height(height)                                  # Fourier-resamples the carried column on a resize
        total, src, a = tr.advance(n_cols)                     # the scalar index recurrence of Online_Linear_2D_resampler.push
        old_in = tr.old_data
        if old_in.dtype != np.float64 or not old_in.flags.c_contiguous:
            old_in = np.ascontiguousarray(old_in, np.float64)
        n_out = len(src)
        pix = np.empty((height, max(n_out, 1)), np.uint32)
        old_out = np.empty(height)
        rc = fr._lib.frt_screen_columns(norm_ptr, freq.size, n_cols, p_freq, p_targets, height, old_in.ctypes.data,
                                        src.ctypes.data if n_out else None, a.ctypes.data if n_out else None, n_out, p_lut,
                                        pix.ctypes.data if n_out else None, old_out.ctypes.data)
        if rc:
            from .. import _lib
            _lib.check(rc)
        tr.old_data = old_out
        if n_out and total == n_out:
            return pix
        out = np.full((height, max(total, 0)), lut[0], np.uint32)              # columns the resampler left at zero map to lut[0]
        if n_out:
            out[:, :n_out] = pix[:, :n_out]
        return out
