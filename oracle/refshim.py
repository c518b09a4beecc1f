"""oracle/refshim.py — import the *unmodified* reference from its checkout (build container only).

The DSP modules of the reference only need two incidental imports stubbed:
`friture.audiobackend` for the constants SAMPLING_RATE / FRAMES_PER_BUFFER (audioproc.py:24,
ringbuffer.py:25) and `PyQt6.QtGui.QColor` for colour packing (color_tranform.py:26,44-46).
Used by oracle/make_golden.py to validate oracle/dsp.py and to record golden fixtures; nothing
that runs on the GPU box may import this module (the checkout does not exist there).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("FRITURE_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "friture"))


def install():
    if not available():
        raise RuntimeError(f"reference checkout not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    ab = types.ModuleType("friture.audiobackend")
    ab.SAMPLING_RATE = 48000
    ab.FRAMES_PER_BUFFER = 512
    sys.modules.setdefault("friture.audiobackend", ab)

    class QColor:
        def __init__(self, r, g, b):
            self._v = 0xFF000000 | (int(r) << 16) | (int(g) << 8) | int(b)

        def rgb(self):
            return self._v

    pyqt = types.ModuleType("PyQt6")
    qtgui = types.ModuleType("PyQt6.QtGui")
    qtgui.QColor = QColor
    pyqt.QtGui = qtgui
    sys.modules.setdefault("PyQt6", pyqt)
    sys.modules.setdefault("PyQt6.QtGui", qtgui)
