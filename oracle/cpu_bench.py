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


def octave_blocks(args):
    """One channel of bench.py's bank legs through the oracle, block by block as the widget feeds it
    (friture/octavespectrum.py:91-122): `blocks` blocks of 1024 samples -> bank -> smoothed band energies -> dB.
    mode "iir": the exact bank (friture/filter.py:86-118) through oracle/iir_ref.c; "ola": the production FFT
    overlap-add bank (friture/octavefilters.py:49-58).  Returns (octave-band units, digest)."""
    channel, bpo, blocks, mode = args
    from oracle import dsp
    t = dsp.load_filter_tables()
    x = synth_prefix(1000 + channel, 1024 * blocks).astype(np.float64)
    alphas, kernels = dsp.band_smoothing_setup(bpo, 1.0)
    prev = [0.0] * (9 * bpo)
    if mode == "iir":
        boct, aoct = list(t[f"boct_{bpo}"]), list(t[f"aoct_{bpo}"])
        zis = dsp.iir_bank_filtic(t["bdec"], t["adec"], boct, aoct)
    else:
        bank = dsp.OlaBank(bpo, t)
    digest = 0.0
    for b in range(blocks):
        blk = x[1024 * b: 1024 * (b + 1)]
        if mode == "iir":
            y, _dec, zis = dsp.iir_bank(t["bdec"], t["adec"], boct, aoct, blk, zis)
        else:
            y, _dec = bank.filter(blk)
        prev = dsp.band_energies(y, kernels, alphas, prev)
        digest += float(dsp.band_db(prev).sum())
    return blocks * 9 * bpo, digest


def gcc_windows(args):
    """`windows` GCC-PHAT window pairs of L samples (friture/signal/correlation.py:24-43) + arg-max, as bench.py's
    gcc_leg synthesises them (delay 37).  Returns (windows, number of windows whose arg-max is 37)."""
    seed, L, windows = args
    from oracle import dsp
    rng = np.random.default_rng(seed)
    hits = 0
    for _ in range(windows):
        d0 = 0.25 * rng.standard_normal(L)
        d1 = np.roll(d0, 37) + 0.025 * rng.standard_normal(L)
        xc, _, _ = dsp.gcc_phat(d0, d1)
        hits += int(np.argmax(np.abs(xc)) == 37)
    return windows, hits
