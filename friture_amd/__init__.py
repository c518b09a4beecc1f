"""friture_amd — MI355X (gfx950) backend for Friture's spectral-analysis hot path.

Host-side mirror of the reference interfaces for that path (same class / function names,
argument meaning and error behaviour as friture/audioproc.py, friture/octavefilters.py,
friture/filter.py, friture/signal/*.py) on top of the C ABI of libfriture_hip.so
(include/friture_hip.h).  There is no CPU fallback: every compute call goes to the HIP library
and raises if it is missing or no gfx950 device is visible.
"""
from .constants import FRAMES_PER_BUFFER, SAMPLING_RATE  # noqa: F401

__version__ = "0.1.0"
