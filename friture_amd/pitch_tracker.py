"""friture/pitch_tracker.py:160-428 on the GPU: `PitchTracker`, `calcCosineKernel`,
`fastParabolicInterp`, plus the batch engine `PitchEngine`.

The tracker matches each frame's log-frequency spectrum against SWIPE-style harmonic kernels; the
table construction below (one-off, host side) produces the same numbers as the reference's loop over
harmonics, the per-frame work — spectrum, log-grid interpolation, the [candidates x grid] product,
peak refinement and the voiced/unvoiced gate — runs in the kernels of csrc/pitch.hip (frt_pitch_*).
The Qt widget around it (PitchTrackerWidget, :57-158) is out of scope.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

from . import _lib
from .audioproc import audioproc
from .constants import SAMPLING_RATE
from .ringbuffer import RingBuffer

# defaults of friture/pitch_tracker_settings.py:27-34
DEFAULT_FFT_SIZE = 4096
DEFAULT_MIN_FREQ = 65
DEFAULT_MAX_FREQ = 1047
DEFAULT_DURATION = 10
DEFAULT_MIN_DB = -50.0
DEFAULT_C_RES = 10
DEFAULT_P_CONF = 0.50
DEFAULT_P_DELTA = 2

_HARMONICS = np.array([1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 13, 17, 19, 23])
_PEAK_WIDTH = 0.15
_VALLEY_WIDTH = 1 - _PEAK_WIDTH


def fastParabolicInterp(y1, y2, y3):
    """Vertex (offset from the centre sample, height) of the parabola through three neighbouring
    pitch strengths (pitch_tracker.py:160-193)."""
    curvature = (y1 - 2 * y2 + y3) / 2
    tilt = (y3 - y1) / 2
    vx = -tilt / (2 * curvature + np.finfo(np.float64).eps)
    return vx, curvature * vx ** 2 + tilt * vx + y2


def calcCosineKernel(f, freqList):
    """Kernel of candidate `f` over the frequency grid (pitch_tracker.py:195-264).

    Around every integer multiple i <= 23 of f, the grid points with -0.85 < freq/f - i < 0.15 carry a
    lobe: a cosine peak of half-width 0.15 (full height at the harmonics in use, a quarter elsewhere)
    preceded by a negative half-height cosine valley.  The intervals of different i do not overlap, so
    every grid point is classified once instead of sweeping the grid per harmonic."""
    freqList = np.asarray(freqList, np.float64)
    in_use = min(int(freqList[-1] / f), len(_HARMONICS))
    used = np.zeros(_HARMONICS[-1] + 2, bool)
    used[_HARMONICS[:in_use]] = True

    ratio = freqList / f
    # the multiple whose interval (-0.85, 0.15) can contain this point, and the distance from it
    mult = np.clip(np.ceil(ratio - _PEAK_WIDTH), 0, _HARMONICS[-1] + 1).astype(int)
    k = np.zeros_like(freqList)
    for cand in (mult, mult + 1):            # ceil() sits on the boundary for points exactly 0.15 above a multiple
        cand = np.minimum(cand, _HARMONICS[-1] + 1)
        a = ratio - cand
        live = (cand >= 1) & (cand <= _HARMONICS[-1])
        sel = used[cand]
        in_peak = live & (np.abs(a) < _PEAK_WIDTH)
        in_valley = live & (-_VALLEY_WIDTH < a) & (a < np.where(sel, -_PEAK_WIDTH, _PEAK_WIDTH)) & ~in_peak
        k[in_valley] = -np.cos((a[in_valley] + 0.5) / ((_VALLEY_WIDTH - _PEAK_WIDTH) / 2) * (np.pi / 2)) / 2
        k[in_peak] = np.cos(a[in_peak] / _PEAK_WIDTH * (np.pi / 2)) / np.where(sel[in_peak], 1, 4)

    knee = f * (2 + _PEAK_WIDTH)             # flat up to harmonic 2.15, then 1/sqrt(freq)
    k *= np.where(freqList <= knee, np.sqrt(1.0 / knee), np.sqrt(1.0 / freqList)) / np.sqrt(1.0 / knee)
    k /= np.sum(k[k > 0])
    k /= in_use / len(_HARMONICS)
    return k


def swipe_tables(sample_rate=SAMPLING_RATE, min_freq=DEFAULT_MIN_FREQ, max_freq=DEFAULT_MAX_FREQ, cres=DEFAULT_C_RES):
    """(log-spaced grid up to Nyquist, candidates = grid below max_freq, kernel matrix) — _init_swipe, :334-355."""
    count = int(np.log2(sample_rate / (2 * min_freq)) * (1200 / cres))
    grid = np.logspace(np.log2(min_freq), np.log2(sample_rate // 2), num=count, base=2)
    candidates = grid[:np.searchsorted(grid, max_freq)]
    kernels = np.zeros((len(candidates), len(grid)))
    for row, f in enumerate(candidates):
        kernels[row] = calcCosineKernel(f, grid)
    return grid, candidates, kernels


class PitchEngine:
    """Batch driver of frt_pitch_*: every complete frame of x[C][T] -> one estimate (NaN = unvoiced)."""

    def __init__(self, fft_size=DEFAULT_FFT_SIZE, hop=None, n_channels=1, sample_rate=SAMPLING_RATE, grid=None, kernels=None,
                 min_db=DEFAULT_MIN_DB, conf=DEFAULT_P_CONF, p_delta=DEFAULT_P_DELTA):
        self._lib = _lib.init()
        if grid is None or kernels is None:
            grid, _, kernels = swipe_tables(sample_rate)
        self.fft_size, self.hop = int(fft_size), int(hop if hop is not None else fft_size // 4)
        self.n_channels = int(n_channels)
        self.grid = np.ascontiguousarray(grid, np.float64)
        kernels = np.ascontiguousarray(kernels, np.float64)
        if kernels.ndim != 2 or kernels.shape[1] != self.grid.size:
            raise ValueError("kernels must be [candidates, len(grid)]")
        self._h = ctypes.c_void_p()
        _lib.check(self._lib.frt_pitch_create(ctypes.byref(self._h), self.fft_size, self.hop, self.n_channels, float(sample_rate),
                                              self.grid.ctypes.data, self.grid.size, kernels.ctypes.data, kernels.shape[0],
                                              float(min_db), float(conf), float(p_delta)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.frt_pitch_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def frames_for(self, n_samples: int) -> int:
        return int(self._lib.frt_pitch_frames_for(self._h, int(n_samples)))

    def reset(self):
        _lib.check(self._lib.frt_pitch_reset(self._h))

    def set_scratch_limit(self, n_bytes: int):
        _lib.check(self._lib.frt_pitch_set_scratch_limit(self._h, int(n_bytes)))

    def set_gate(self, min_db, conf, p_delta):
        _lib.check(self._lib.frt_pitch_set_gate(self._h, float(min_db), float(conf), float(p_delta)))

    @property
    def previous(self):
        p = np.empty(self.n_channels, np.float64)
        _lib.check(self._lib.frt_pitch_get_previous(self._h, p.ctypes.data))
        return p

    @previous.setter
    def previous(self, values):
        p = np.ascontiguousarray(np.broadcast_to(np.asarray(values, np.float64), (self.n_channels,)))
        _lib.check(self._lib.frt_pitch_set_previous(self._h, p.ctypes.data))

    def track(self, x, with_raw: bool = False):
        """x: [C, T] float64 (numpy, or a torch CUDA tensor: the result stays in HBM).
        Returns f0 [C, F]; with_raw adds [3, C, F] = (estimate before gating, confidence, dBFS)."""
        if type(x).__module__.startswith("torch"):
            import torch
            if not (x.is_cuda and x.dtype == torch.float64 and x.is_contiguous()):
                raise ValueError("expected a contiguous float64 CUDA tensor")
            if x.dim() == 1:
                x = x[None, :]
            if x.shape[0] != self.n_channels:
                raise ValueError(f"expected {self.n_channels} channels, got {x.shape[0]}")
            F = self.frames_for(x.shape[1])
            f0 = torch.empty((self.n_channels, F), dtype=torch.float64, device=x.device)
            raw = torch.empty((3, self.n_channels, F), dtype=torch.float64, device=x.device) if with_raw else None
            _lib.check(self._lib.frt_pitch_set_stream(self._h, ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
            if F:
                _lib.check(self._lib.frt_pitch_track(self._h, ctypes.c_void_p(x.data_ptr()), x.shape[1], x.stride(0),
                                                     ctypes.c_void_p(f0.data_ptr()),
                                                     ctypes.c_void_p(raw.data_ptr()) if with_raw else None, None))
            return (f0, raw) if with_raw else f0
        x = np.ascontiguousarray(x, np.float64)
        if x.ndim == 1:
            x = x[None, :]
        if x.shape[0] != self.n_channels:
            raise ValueError(f"expected {self.n_channels} channels, got {x.shape[0]}")
        F = self.frames_for(x.shape[1])
        f0 = np.empty((self.n_channels, F), np.float64)
        raw = np.empty((3, self.n_channels, F), np.float64) if with_raw else None
        if F:
            _lib.check(self._lib.frt_pitch_track(self._h, x.ctypes.data, x.shape[1], x.shape[1], f0.ctypes.data,
                                                 raw.ctypes.data if with_raw else None, None))
        return (f0, raw) if with_raw else f0


class PitchTracker:
    """Streaming tracker over a ring buffer, with the reference's interface (pitch_tracker.py:266-428)."""

    def __init__(self, input_buf: RingBuffer, fft_size: int = DEFAULT_FFT_SIZE, overlap: float = 0.75,
                 sample_rate: int = SAMPLING_RATE, min_freq: float = DEFAULT_MIN_FREQ, max_freq: float = DEFAULT_MAX_FREQ,
                 min_db: float = DEFAULT_MIN_DB, cres: int = DEFAULT_C_RES, conf: float = DEFAULT_P_CONF,
                 p_delta: int = DEFAULT_P_DELTA):
        self.fft_size = fft_size
        self.overlap = overlap
        self.sample_rate = sample_rate
        self.min_freq = min_freq
        self.max_freq = max_freq
        self.min_db = min_db
        self.cres = cres
        self.conf = conf
        self.p_delta = p_delta
        self.prev_f0 = None

        self.input_buf = input_buf
        self.input_buf.grow_if_needed(fft_size)
        self.next_in_offset = self.input_buf.offset

        self.out_buf = RingBuffer()
        self.out_offset = self.out_buf.offset

        self.proc = audioproc()
        self.proc.set_fftsize(self.fft_size)
        self._engine = None
        self._init_swipe()

    def set_input_buffer(self, new_buf: RingBuffer) -> None:
        self.input_buf = new_buf
        self.input_buf.grow_if_needed(self.fft_size)
        self.next_in_offset = self.input_buf.offset

    def _init_swipe(self):
        self.logSpacedFreqs, self.pitchCandidates, self.kernels = swipe_tables(self.sample_rate, self.min_freq, self.max_freq,
                                                                               self.cres)
        self._engine = None                   # tables changed: the device plan is rebuilt at the next estimate

    def _step(self) -> int:
        return math.floor(self.fft_size * (1.0 - self.overlap))

    def _plan(self) -> PitchEngine:
        if self._engine is None:
            self._engine = PitchEngine(self.fft_size, max(1, self._step()), 1, self.sample_rate, self.logSpacedFreqs, self.kernels,
                                       self.min_db, self.conf, self.p_delta)
        # thresholds and the previous estimate are plain attributes upstream (the widget writes them, :142-146)
        self._engine.set_gate(self.min_db, self.conf, self.p_delta)
        self._engine.previous = np.nan if self.prev_f0 is None else self.prev_f0
        return self._engine

    def _run(self, samples):
        """All complete frames of one contiguous float64 run ([T] or [rows, T]) -> estimates; keeps prev_f0 in step.
        The spectrum is row 0's; the level of the gate is the RMS over EVERY row of the frame (pitch_tracker.py:404-407:
        `np.sqrt(np.mean(frame**2))` on the 2-D frame, two rows when the shared ring buffer is in dual-channel mode)."""
        samples = np.asarray(samples, np.float64)
        if samples.ndim == 1 or samples.shape[0] == 1:
            est = self._plan().track(np.ascontiguousarray(samples.reshape(-1)))[0]
        else:
            # dual-channel frames: estimate, confidence on the device from row 0; the gate (one comparison chain per frame,
            # sequential in prev_f0) on the host with the all-rows level
            eng = self._plan()
            _, raw = eng.track(np.ascontiguousarray(samples[0]), with_raw=True)
            f0_raw, conf = raw[0, 0], raw[1, 0]
            step, n = max(1, self._step()), self.fft_size
            csum = np.concatenate([[0.0], np.cumsum(np.sum(samples * samples, axis=0))])
            est = np.empty_like(f0_raw)
            prev = self.prev_f0
            for f in range(f0_raw.size):
                rms = np.sqrt((csum[f * step + n] - csum[f * step]) / (samples.shape[0] * n))
                dbfs = 20 * np.log10(rms + np.finfo(np.float64).eps)
                f0 = f0_raw[f]
                jump = 12 * np.abs(np.log2(f0 / prev)) if prev is not None else 0
                if (dbfs < self.min_db) or (conf[f] < self.conf) or (jump > self.p_delta) or np.isnan(f0):
                    prev, est[f] = None, np.nan
                else:
                    prev, est[f] = float(f0), f0
        if est.size:
            self.prev_f0 = None if np.isnan(est[-1]) else float(est[-1])
        return est

    def estimate_pitch(self, frame: np.ndarray):
        """frame: [rows, fft_size]: spectrum from row 0, level from every row.  Hz, or nan if unvoiced."""
        frame = np.asarray(frame, np.float64)
        if frame.shape[-1] != self.fft_size:
            raise ValueError(f"estimate_pitch expects {self.fft_size} samples, got {frame.shape[-1]}")
        return float(self._run(frame)[0])

    def new_frames(self):
        assert self.input_buf.offset >= self.next_in_offset
        while self.next_in_offset + self.fft_size <= self.input_buf.offset:
            yield self.input_buf.data_indexed(self.next_in_offset + self.fft_size, self.fft_size)
            self.next_in_offset += self._step()

    def update(self) -> bool:
        """Estimates for every frame completed since the last call, as ONE batch: consecutive frames are
        a contiguous run of the ring (the reference estimates them one by one, :313-317)."""
        assert self.input_buf.offset >= self.next_in_offset
        step = self._step()
        avail = self.input_buf.offset - self.next_in_offset
        count = 0 if avail < self.fft_size else (avail - self.fft_size) // step + 1 if step > 0 else 0
        new = []
        if count:
            span = self.fft_size + (count - 1) * step
            run = self.input_buf.data_indexed(self.next_in_offset + span, span)
            new = list(self._run(np.ascontiguousarray(run)))
            self.next_in_offset += count * step
        self.out_buf.push(np.array([new]), 0)
        self.out_offset = self.out_buf.offset
        return len(new) != 0

    def get_estimates(self, time_s: float) -> np.ndarray:
        num_results = math.floor(time_s / (self._step() / self.sample_rate)) + 1
        return self.out_buf.data_indexed(self.out_offset, num_results)[0, :]

    def get_latest_estimate(self) -> float:
        return self.out_buf.data_indexed(self.out_offset, 1)[0, 0]
