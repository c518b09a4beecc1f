"""friture/signal/color_tranform.py:28-51 on the GPU (the module keeps the reference's file name,
typo included): clip to [0, 1], truncate to a 256-entry index, look the colour word up."""
from __future__ import annotations

import numpy as np

from .. import _lib, palette


class Color_Transform:
    def __init__(self) -> None:
        self._lib = _lib.init()
        self.prepare_palette()

    def prepare_palette(self):
        self.colors = palette.cmr_lut()

    def push(self, data):
        data = np.ascontiguousarray(data, np.float64)
        out = np.empty(data.shape, np.uint32)
        lut = np.ascontiguousarray(self.colors, np.uint32)
        _lib.check(self._lib.frt_colour_map(lut.ctypes.data, data.ctypes.data, data.size, out.ctypes.data))
        return out
