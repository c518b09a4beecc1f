"""ctypes binding of libfriture_hip.so (include/friture_hip.h).

The library is built in-tree by `python -m friture_amd.build`.  Loading fails loudly when it is
missing; there is no fallback implementation.
"""
from __future__ import annotations

import ctypes
import os
import sys
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_uint32, c_void_p
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libfriture_hip.so"

FRT_STFT_PSD, FRT_STFT_DB, FRT_STFT_NORM, FRT_STFT_IMAGE = 0, 1, 2, 3


class DelayReadout(ctypes.Structure):
    """frt_delay_readout of include/friture_hip.h"""
    _fields_ = [("argmax", c_int), ("correlation_pct", c_int), ("delay_ms", c_double), ("distance_m", c_double),
                ("extremum", c_double)]


class FritureHipError(RuntimeError):
    """A C-ABI call returned a negative status (the message is frt_last_error())."""

    def __init__(self, status: int, message: str):
        super().__init__(f"libfriture_hip status {status}: {message}")
        self.status = status


# name -> (restype, argtypes); kept in one table so that tests can check the export list
SIGNATURES = {
    "frt_init": (c_int, [c_int, POINTER(c_int), POINTER(c_int64)]),
    "frt_device_properties": (c_int, [c_int, POINTER(c_int), POINTER(c_int64)]),
    "frt_last_error": (c_char_p, []),
    "frt_version": (c_char_p, []),
    "frt_is_device_pointer": (c_int, [c_void_p]),
    "frt_set_option": (c_int, [c_char_p, c_int]),
    "frt_get_option": (c_int, [c_char_p, POINTER(c_int)]),
    "frt_stft_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int]),
    "frt_stft_destroy": (None, [c_void_p]),
    "frt_stft_set_stream": (c_int, [c_void_p, c_void_p]),
    "frt_stft_set_epilogue": (c_int, [c_void_p, POINTER(c_double), c_double, c_double, POINTER(c_uint32)]),
    "frt_stft_run": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int64, c_void_p, POINTER(c_int64)]),
    "frt_stft_run_split": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int64, c_void_p, c_void_p, POINTER(c_int64)]),
    "frt_stft_psd": (c_int, [c_void_p, POINTER(c_float), c_int64, POINTER(c_float), POINTER(c_int64)]),
    "frt_stft_image": (c_int, [c_void_p, POINTER(c_float), c_int64, POINTER(c_uint32), POINTER(c_int64)]),
    "frt_stft_analyzelive_f64": (c_int, [c_void_p, POINTER(c_double), POINTER(c_double)]),
    "frt_stft_frames_for": (c_int64, [c_void_p, c_int64]),
    "frt_stft_set_run_length": (c_int, [c_void_p, c_int]),
    "frt_octbank_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int] + [POINTER(c_double)] * 6),
    "frt_octbank_destroy": (None, [c_void_p]),
    "frt_octbank_set_stream": (c_int, [c_void_p, c_void_p]),
    "frt_octbank_reset": (c_int, [c_void_p]),
    "frt_octbank_set_chunk": (c_int, [c_void_p, c_int]),
    "frt_octbank_packed_length": (c_int64, [c_void_p, c_int]),
    "frt_octbank_filter": (c_int, [c_void_p, c_void_p, c_int, c_void_p, POINTER(c_int)]),
    "frt_octbank_state_length": (c_int, [c_void_p]),
    "frt_octbank_get_state": (c_int, [c_void_p, POINTER(c_double)]),
    "frt_octbank_set_state": (c_int, [c_void_p, POINTER(c_double)]),
    "frt_octbank_energies": (c_int, [c_void_p, c_void_p, c_int64, c_int, POINTER(c_double), POINTER(c_double), c_int,
                                     c_void_p]),
    "frt_decimate_multiple": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, POINTER(c_int)]),
    "frt_decimate_multiple_state": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, POINTER(c_int), c_void_p]),
    "frt_gcc_create": (c_int, [POINTER(c_void_p), c_int, c_int]),
    "frt_gcc_destroy": (None, [c_void_p]),
    "frt_gcc_set_stream": (c_int, [c_void_p, c_void_p]),
    "frt_gcc_phat": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "frt_gcc_readout": (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_double, c_double, c_void_p, c_void_p]),
    "frt_freq_resample": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "frt_time_resample": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "frt_fourier_resample": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int]),
    "frt_specgram_create": (c_int, [POINTER(c_void_p), c_int, c_double, c_int]),
    "frt_specgram_destroy": (None, [c_void_p]),
    "frt_specgram_set_epilogue": (c_int, [c_void_p, POINTER(c_double), c_double, c_double, POINTER(c_uint32)]),
    "frt_specgram_set_screen": (c_int, [c_void_p, POINTER(c_double), POINTER(c_double), c_int]),
    "frt_specgram_set_ratio": (c_int, [c_void_p, c_double, c_double]),
    "frt_specgram_push": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, POINTER(c_int), POINTER(c_int)]),
    "frt_specgram_reset": (c_int, [c_void_p]),
    "frt_colour_map": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "frt_screen_columns": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                   c_void_p, c_void_p]),
    "frt_exp_smooth_2d": (c_int, [c_void_p, c_int, c_double, c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p]),
    "frt_exp_smooth_groups": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "frt_spectrum_post": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_int, c_double, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, POINTER(c_int), POINTER(c_int)]),
    "frt_pitch_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_double, c_void_p, c_int, c_void_p, c_int, c_double,
                                 c_double, c_double]),
    "frt_pitch_destroy": (None, [c_void_p]),
    "frt_pitch_set_stream": (c_int, [c_void_p, c_void_p]),
    "frt_pitch_reset": (c_int, [c_void_p]),
    "frt_pitch_set_previous": (c_int, [c_void_p, c_void_p]),
    "frt_pitch_get_previous": (c_int, [c_void_p, c_void_p]),
    "frt_pitch_set_gate": (c_int, [c_void_p, c_double, c_double, c_double]),
    "frt_pitch_set_scratch_limit": (c_int, [c_void_p, c_int64]),
    "frt_pitch_frames_for": (c_int64, [c_void_p, c_int64]),
    "frt_pitch_track": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, POINTER(c_int64)]),
    "frt_delay_create": (c_int, [POINTER(c_void_p), POINTER(c_double), POINTER(c_double), c_int, c_int]),
    "frt_delay_destroy": (None, [c_void_p]),
    "frt_delay_stream": (c_void_p, [c_void_p]),
    "frt_delay_push": (c_int, [c_void_p, c_void_p, c_int, POINTER(c_int64)]),
    "frt_delay_reserve": (c_int, [c_void_p, c_int]),
    "frt_delay_window": (c_int, [c_void_p, c_int64, c_int, POINTER(c_void_p), POINTER(c_void_p)]),
    "frt_delay_window_std": (c_int, [c_void_p, c_void_p, c_void_p, c_int, POINTER(c_double)]),
    "frt_delay_demean": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "frt_lfilter_f64": (c_int, [POINTER(c_double), POINTER(c_double), c_int, POINTER(c_double), c_int, POINTER(c_double),
                                POINTER(c_double), POINTER(c_double)]),
}

_lib = None
_initialised = False
bound_device = 0          # the HIP device frt_init bound this process to (set by init)


def load() -> ctypes.CDLL:
    """dlopen the library (no GPU needed) and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m friture_amd.build` "
                          "(friture_amd has no CPU fallback)")
    # If torch is (going to be) used in this process its bundled HIP runtime must be the one the
    # library binds to, otherwise device pointers would belong to a different runtime instance.
    if "torch" not in sys.modules and os.environ.get("FRITURE_AMD_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = ctypes.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != 0:
        raise FritureHipError(status, load().frt_last_error().decode(errors="replace"))


def init(device: int | None = None) -> ctypes.CDLL:
    """Load the library and bind the process to one gfx950 device (LOCAL_RANK by default)."""
    global _initialised
    lib = load()
    if not _initialised:
        if device is None:
            # one process per GPU: the launcher's LOCAL_RANK decides; without it, the device torch was told to use
            # (torch.cuda.current_device() is 0 until set_device is called, so it must not override LOCAL_RANK)
            if "LOCAL_RANK" in os.environ:
                device = int(os.environ["LOCAL_RANK"])
            else:
                device = 0
                if "torch" in sys.modules:
                    import torch
                    if torch.cuda.is_available() and torch.cuda.is_initialized():
                        device = torch.cuda.current_device()
        check(lib.frt_init(device, None, None))
        global bound_device
        bound_device = int(device)
        _initialised = True
    return lib


def set_option(name: str, value: int) -> None:
    """frt_set_option: force one of two product code paths of an entry point (tests, A/B runs); value < 0 restores the
    shape rule.  The library never reads the environment."""
    check(load().frt_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    v = c_int(0)
    check(load().frt_get_option(name.encode(), ctypes.byref(v)))
    return v.value


def device_info(device: int = 0) -> tuple[int, int]:
    lib = load()
    ncu, hbm = c_int(0), c_int64(0)
    check(lib.frt_device_properties(device, ctypes.byref(ncu), ctypes.byref(hbm)))      # does not rebind the process
    return ncu.value, hbm.value
