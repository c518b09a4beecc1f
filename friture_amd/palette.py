"""Colour palette of the spectrogram: the monochrome-compatible "CMR" map packed as 0xFFRRGGBB.

The reference ships the map as a generated table (friture/plotting/generated_cmrmap.py, produced
by friture/plotting/cmrmap_generate.py:57-83: cubic splines through nine anchor colours, shifted
and scaled to [0, 1]) and packs it with QColor(int(r*255), int(g*255), int(b*255)).rgb()
(friture/signal/color_tranform.py:36-46).  The construction is deterministic, so the 256 colour
words are rebuilt here from the anchors; tests/test_oracle_golden.py::test_image pins them to the
reference's words.
"""
from __future__ import annotations

import functools

import numpy as np

# black - purple - red - yellow - white anchors (Rappaport 2002, adjusted for linear luminance)
_ANCHORS = np.array([
    [0.0, 0.0, 0.0], [0.1, 0.1, 0.35], [0.3, 0.15, 0.65], [0.6, 0.2, 0.50], [1.0, 0.25, 0.15],
    [0.9, 0.55, 0.0], [0.9, 0.75, 0.1], [0.9, 0.9, 0.5], [1.0, 1.0, 1.0]])


@functools.lru_cache(maxsize=None)
def cmr_colours(n: int = 256) -> np.ndarray:
    """n x 3 float64 RGB in [0, 1]."""
    from scipy.interpolate import splev, splrep
    knots = np.linspace(0.0, 1.0, len(_ANCHORS))
    grid = np.linspace(0.0, 1.0, n)
    rgb = np.empty((n, 3))
    for ch in range(3):
        rgb[:, ch] = splev(grid, splrep(knots, _ANCHORS[:, ch], s=0))
    rgb -= rgb.min()
    rgb /= rgb.max()
    return rgb


def pack_rgb32(rgb: np.ndarray) -> np.ndarray:
    """0xFF000000 | R << 16 | G << 8 | B with truncating 8-bit quantisation."""
    q = (np.asarray(rgb) * 255).astype(np.int64)
    return (0xFF000000 | (q[:, 0] << 16) | (q[:, 1] << 8) | q[:, 2]).astype(np.uint32)


def cmr_lut() -> np.ndarray:
    """The 256 colour words Color_Transform.prepare_palette builds."""
    return pack_rgb32(cmr_colours(256))
