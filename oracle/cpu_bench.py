"""oracle/cpu_bench.py — worker of bench.py's multi-core CPU baseline (TEST/BENCH INFRASTRUCTURE).

Each worker process regenerates the same synthetic channel prefix and runs the numpy oracle of the
spectrogram path over it `passes` times; bench.py times the whole pool.  Imports numpy and
oracle.dsp only, so that `spawn`ed workers start quickly and never touch the GPU runtime.
"""
import numpy as np


def synth_prefix(channel: int, n: int) -> np.ndarray:
    """The first n samples of bench.py's synth_channel(channel, ...): the generator is sequential, so a
    shorter draw from the same seed is a prefix of the longer one."""
    return (0.25 * np.random.default_rng(42 + channel).standard_normal(n, dtype=np.float32)).astype(np.float32)


def spectrogram_passes(args):
    channel, n_fft, hop, frames, passes, weight, lut = args
    from oracle import dsp
    x = synth_prefix(channel, n_fft + hop * (frames - 1)).astype(np.float64)
    digest = 0.0
    for _ in range(passes):
        img = dsp.spectrogram_image(x, n_fft, hop, weight, -140.0, 0.0, lut)
        digest += float(img[::97, ::31].sum())
    return frames * passes, digest


_barrier = None


def init_worker(barrier):
    """Pool initializer: keep the start barrier and import the oracle before any timed work."""
    global _barrier
    _barrier = barrier
    from oracle import dsp  # noqa: F401


def wait_ready(_):
    """First job of every worker: returns once all workers of the pool are up (so that the timed jobs
    do not include interpreter start-up of late workers)."""
    _barrier.wait(timeout=120)
    return True
