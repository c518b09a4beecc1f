"""oracle/make_golden_pitch.py — record tests/golden/pitch.npz from the UNMODIFIED reference
(friture/pitch_tracker.py executed through oracle/refshim.py with its Qt/UI imports stubbed) and
check oracle/dsp.py's restatement against it.  Build container only (needs /root/reference).

    python -m oracle.make_golden_pitch
"""
import sys
import types
from pathlib import Path

import numpy as np

from oracle import dsp, refshim

GOLD = Path(__file__).resolve().parents[1] / "tests" / "golden"


def import_reference_pitch_tracker():
    """friture.pitch_tracker pulls in the Qt widget stack at import time (pitch_tracker.py:31-55);
    only PitchTracker / calcCosineKernel / fastParabolicInterp are needed, so every UI module gets
    a stand-in whose attributes are empty classes."""
    refshim.install()

    class Blank:
        def __init__(self, *a, **k):
            pass

    def stand_in(name):
        m = types.ModuleType(name)
        m.__getattr__ = lambda attr: type(attr, (Blank,), {})
        sys.modules[name] = m
        return m

    for name in ("PyQt6.QtWidgets", "PyQt6.QtCore", "friture.audiobuffer", "friture.curve", "friture.pitch_tracker_data",
                 "friture.store", "friture.plotting.coordinateTransform"):
        stand_in(name)
    import PyQt6
    PyQt6.QtWidgets = sys.modules["PyQt6.QtWidgets"]
    PyQt6.QtCore = sys.modules["PyQt6.QtCore"]
    import friture.pitch_tracker as pt
    return pt


def voiced(seed, n, f0_path, fs=48000.0, harmonics=(1.0, 0.8, 0.5, 0.35, 0.2, 0.1), noise=1e-3):
    """Harmonic tone following f0_path (Hz per sample) + a little noise, float32 PCM."""
    phase = 2 * np.pi * np.cumsum(f0_path) / fs
    x = sum(a * np.sin((h + 1) * phase) for h, a in enumerate(harmonics))
    x = 0.2 * x / np.max(np.abs(x)) + noise * np.random.default_rng(seed).standard_normal(n)
    return x.astype(np.float32)


def signals(n_fft):
    n = n_fft * 12
    t = np.arange(n)
    return {
        "steady220": voiced(1, n, np.full(n, 220.0)),
        "glide": voiced(2, n, 110.0 * 2 ** (t / n)),                                       # one octave up, smooth
        "jump": voiced(3, n, np.where(t < n // 2, 196.0, 392.0)),                          # octave jump: p_delta gate
        "quiet": (voiced(4, n, np.full(n, 330.0)) * 1e-3).astype(np.float32),              # below min_db
        "noise": (0.25 * np.random.default_rng(5).standard_normal(n)).astype(np.float32),  # low confidence
        "silence": np.zeros(n, np.float32),
        "high900": voiced(6, n, np.full(n, 900.0), harmonics=(1.0, 0.5, 0.25)),
    }


def main():
    pt = import_reference_pitch_tracker()
    from friture.ringbuffer import RingBuffer
    out = {}
    worst = 0.0
    for n_fft, overlap in ((4096, 0.75), (2048, 0.5), (1024, 0.5)):
        hop = int(np.floor(n_fft * (1.0 - overlap)))
        tracker = pt.PitchTracker(RingBuffer(), fft_size=n_fft, overlap=overlap)
        freqs, kernels = dsp.swipe_tables()
        assert np.array_equal(freqs, tracker.logSpacedFreqs), "log grid differs from the reference"
        assert np.array_equal(kernels, tracker.kernels), "kernel table differs from the reference"
        if n_fft == 4096:
            out["freqs"] = freqs
            out["kernel_rows"] = np.array([0, 1, 57, 240, 479, 480])
            out["kernel_sample"] = tracker.kernels[out["kernel_rows"]]
            out["kernel_sha_sum"] = np.array([tracker.kernels.sum(), np.abs(tracker.kernels).sum(), (tracker.kernels ** 2).sum()])
        for name, x in signals(n_fft).items():
            tracker.prev_f0 = None
            frames = (len(x) - n_fft) // hop + 1
            ref = np.zeros((2, frames))
            strengths0 = None
            for g in range(frames):
                frame = x[g * hop:g * hop + n_fft].astype(np.float64)[None, :]
                # raw candidate: estimate with the gate opened (fresh tracker state is restored after)
                saved = (tracker.prev_f0, tracker.min_db, tracker.conf, tracker.p_delta)
                tracker.prev_f0, tracker.min_db, tracker.conf, tracker.p_delta = None, -1e9, -1e9, 1e9
                ref[1, g] = tracker.estimate_pitch(frame)
                tracker.prev_f0, tracker.min_db, tracker.conf, tracker.p_delta = saved
                ref[0, g] = tracker.estimate_pitch(frame)
            mine = dsp.pitch_track(x, n_fft, hop, freqs, kernels)
            assert np.array_equal(np.isnan(mine[0]), np.isnan(ref[0])), (n_fft, name)
            err = np.nanmax(np.abs(mine[:2] / ref - 1.0)) if np.any(~np.isnan(ref)) else 0.0
            worst = max(worst, float(np.nan_to_num(err)))
            key = f"N{n_fft}_{name}"
            out[key + "_x"] = x
            out[key + "_f0"] = ref[0]
            out[key + "_raw"] = ref[1]
            out[key + "_conf"] = mine[2]
            out[key + "_dbfs"] = mine[3]
    # the two upstream known answers, as the reference computes them today (both nan: see oracle/dsp.py T1)
    tr = pt.PitchTracker(RingBuffer(), fft_size=32, overlap=0.5)
    kat = np.fft.irfft([0, 0, .5, 0, .7, 0, .4, 0, .2, 0, 0, 0, 0, 0, 0, 0, 0])
    out["kat32_frame"] = kat
    out["kat32_reference_today"] = np.array([tr.estimate_pitch(np.array([kat]))])
    tr.min_db, tr.conf, tr.prev_f0 = -1e9, -1e9, None
    out["kat32_raw"] = np.array([tr.estimate_pitch(np.array([kat]))])
    print(f"oracle vs reference pitch tracker: worst relative difference {worst:.3e}")
    assert worst < 1e-12
    np.savez_compressed(GOLD / "pitch.npz", **out)
    print("wrote", GOLD / "pitch.npz", {k: v.shape for k, v in out.items() if not k.endswith("_x")})


if __name__ == "__main__":
    main()
