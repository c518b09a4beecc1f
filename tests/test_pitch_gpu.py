"""T1 parity: the HIP pitch tracker (csrc/pitch.hip behind frt_pitch_*) against golden vectors recorded
from the reference's PitchTracker and against the oracle.

Tolerances: everything is float64.  The spectrum, the log-grid interpolation and the [481 x 1023]
product differ from numpy only in summation order, so strengths agree to ~1e-13 relative; the parabolic
vertex divides by the curvature of three neighbouring strengths, which amplifies that to ~1e-10 on the
sub-bin offset: estimates are held to 1e-9 relative, confidence to 1e-11, level to 1e-10 dB.  The
voiced/unvoiced pattern must be identical.
"""
import numpy as np
import pytest

from oracle import dsp

pytestmark = pytest.mark.gpu

CASES = [(4096, 1024), (2048, 1024), (1024, 512)]
SIGNALS = ["steady220", "glide", "jump", "quiet", "noise", "silence", "high900"]
TOL_F0, TOL_CONF, TOL_DB = 1e-9, 1e-11, 1e-10


@pytest.fixture(scope="module")
def pt(hip):
    from friture_amd import pitch_tracker
    return pitch_tracker


def close(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if not np.array_equal(np.isnan(a), np.isnan(b)):
        return False
    m = ~np.isnan(a)
    return bool(np.all(np.abs(a[m] - b[m]) <= tol * np.maximum(1.0, np.abs(b[m]))))


@pytest.mark.parametrize("n_fft,hop", CASES)
def test_against_reference_golden(golden, pt, n_fft, hop):
    g = golden("pitch")
    for name in SIGNALS:
        key = f"N{n_fft}_{name}"
        eng = pt.PitchEngine(n_fft, hop)
        f0, raw = eng.track(g[key + "_x"].astype(np.float64), with_raw=True)
        assert f0.shape == (1, len(g[key + "_f0"]))
        assert close(f0[0], g[key + "_f0"], TOL_F0), (key, f0[0], g[key + "_f0"])
        if name != "noise":      # on noise the arg-max sits among near-ties: the raw estimate is not a stable quantity
            assert close(raw[0, 0], g[key + "_raw"], TOL_F0), key
        assert close(raw[1, 0], g[key + "_conf"], TOL_CONF), key
        assert np.all(np.abs(raw[2, 0] - g[key + "_dbfs"]) <= TOL_DB), key


def test_tables_match_reference(golden, pt):
    g = golden("pitch")
    grid, cand, kernels = pt.swipe_tables()
    assert np.array_equal(grid, g["freqs"]) and len(cand) == 481
    assert np.array_equal(kernels[g["kernel_rows"]], g["kernel_sample"])
    vx, vy = pt.fastParabolicInterp(1.0, 3.0, 2.0)
    assert (vx, vy) == dsp.parabolic_vertex(1.0, 3.0, 2.0)


def test_batch_channels_state_and_device_path(golden, pt):
    """Several channels per launch, state carried across calls, torch-resident input."""
    import torch
    g = golden("pitch")
    n_fft, hop = 2048, 512
    names = ["steady220", "glide", "jump", "high900"]
    x = np.stack([g[f"N2048_{n}_x"].astype(np.float64) for n in names])
    freqs, kernels = dsp.swipe_tables()
    ref = [dsp.pitch_track(x[c], n_fft, hop, freqs, kernels) for c in range(len(names))]
    eng = pt.PitchEngine(n_fft, hop, len(names))
    f0, raw = eng.track(x, with_raw=True)
    for c in range(len(names)):
        assert close(f0[c], ref[c][0], TOL_F0), names[c]
        assert close(raw[1, c], ref[c][2], TOL_CONF)
    # two calls split on a frame boundary == one call (the gate's previous estimate lives in the handle)
    eng.reset()
    split = 17
    first = eng.track(x[:, :n_fft + hop * (split - 1)])
    second = eng.track(x[:, hop * split:])
    assert np.array_equal(np.concatenate([first, second], axis=1), f0, equal_nan=True)
    # device-resident input: same bits, result stays in HBM
    eng.reset()
    xd = torch.from_numpy(x).cuda()
    f0d = eng.track(xd)
    assert f0d.is_cuda and np.array_equal(f0d.cpu().numpy(), f0, equal_nan=True)
    # the jump case exercises the p_delta gate: at least one frame is rejected only because of the jump
    j = names.index("jump")
    rejected = np.isnan(f0[j]) & (raw[1, j] >= 0.5) & (raw[2, j] >= -50.0)
    assert rejected.any()


def test_long_run_crosses_scratch_chunks(pt):
    """More frames than one scratch chunk holds (16 KB of scratch per frame at N = 1024, limit set to 256 MB)."""
    n_fft, hop, frames = 1024, 256, 20000
    n = n_fft + hop * (frames - 1)
    t = np.arange(n)
    f_path = 150.0 * 2 ** (1.5 * t / n)
    phase = 2 * np.pi * np.cumsum(f_path) / 48000.0
    x = 0.2 * (np.sin(phase) + 0.6 * np.sin(2 * phase) + 0.3 * np.sin(3 * phase)) + 1e-3 * np.random.default_rng(0).standard_normal(n)
    eng = pt.PitchEngine(n_fft, hop)
    eng.set_scratch_limit(256 << 20)
    f0, raw = eng.track(x, with_raw=True)
    assert f0.shape == (1, frames)
    whole = pt.PitchEngine(n_fft, hop).track(x)                  # default limit: one chunk
    assert np.array_equal(whole, f0, equal_nan=True)
    freqs, kernels = dsp.swipe_tables()
    window = dsp.hann_symmetric(n_fft)
    pick = np.unique(np.concatenate([np.arange(0, 40), np.arange(16350, 16420), np.arange(frames - 40, frames),
                                     np.random.default_rng(1).integers(0, frames, 300)]))
    for f in pick:
        r0, c, db = dsp.pitch_candidate(x[f * hop:f * hop + n_fft], window, freqs, kernels)
        assert abs(raw[0, 0, f] - r0) <= TOL_F0 * r0 and abs(raw[1, 0, f] - c) <= TOL_CONF and abs(raw[2, 0, f] - db) <= TOL_DB, f
    # the gate over the whole run, replayed on the host from the device's own per-frame values
    gate = dsp.PitchGate()
    want = np.array([gate.step(raw[0, 0, f], raw[1, 0, f], raw[2, 0, f]) for f in range(frames)])
    assert np.array_equal(want, f0[0], equal_nan=True)


def test_edge_cases(pt):
    eng = pt.PitchEngine(1024, 512)
    assert eng.track(np.zeros(1023)).shape == (1, 0)
    f0, raw = eng.track(np.zeros(4096), with_raw=True)            # silence: nan spectrum, index 0, gated out by level
    assert np.all(np.isnan(f0)) and np.all(raw[0] == eng.grid[0]) and np.all(np.isnan(raw[1]))
    assert np.allclose(raw[2], 20 * np.log10(np.finfo(np.float64).eps))
    with pytest.raises(ValueError):
        eng.track(np.zeros((2, 4096)))
    from friture_amd._lib import FritureHipError
    with pytest.raises(FritureHipError):
        pt.PitchEngine(1000, 250)                                  # not a power of two
    with pytest.raises(FritureHipError):
        pt.PitchEngine(1024, 256, grid=np.array([3.0, 2.0, 1.0]), kernels=np.zeros((1, 3)))   # grid must increase


def test_streaming_tracker_mirror(golden, pt):
    """PitchTracker over a ring buffer, fed in uneven pushes (upstream's update/new_frames protocol,
    friture/test/test_pitch_tracker.py:27-40,54-71)."""
    from friture_amd.ringbuffer import RingBuffer
    g = golden("pitch")
    buf = RingBuffer()
    tr = pt.PitchTracker(buf, fft_size=4, overlap=0.5)
    buf.push(np.array([np.arange(2)]))
    buf.push(np.array([np.arange(2, 5)]))
    assert [f.tolist() for f in tr.new_frames()] == [[[0, 1, 2, 3]]]
    buf.push(np.array([np.arange(5, 8)]))
    assert [f.tolist() for f in tr.new_frames()] == [[[2, 3, 4, 5]], [[4, 5, 6, 7]]]

    x = g["N4096_jump_x"].astype(np.float64)
    buf = RingBuffer()
    tr = pt.PitchTracker(buf)                                      # defaults: 4096 points, 75 % overlap
    assert not tr.update()
    pos, got_any = 0, False
    for size in [512, 512, 3000, 777, 2048, 1, 3000, 1024, 1023, 2500] * 2:      # the ring holds 10000 samples: keep pushes small
        buf.push(x[None, pos:pos + size])
        pos += size
        got_any |= tr.update()
        assert not tr.update()                                    # nothing new until the next push
    assert got_any
    frames = (pos - 4096) // 1024 + 1
    want = g["N4096_jump_f0"][:frames]
    got = tr.get_estimates((frames - 1) * 1024 / 48000.0)
    assert got.shape == (frames,) and close(got, want, TOL_F0)
    last = tr.get_latest_estimate()
    assert (np.isnan(last) and np.isnan(want[-1])) or abs(last - want[-1]) <= TOL_F0 * want[-1]
    # one frame at a time through estimate_pitch, thresholds changed on the fly like the widget does
    tr2 = pt.PitchTracker(RingBuffer())
    tr2.conf = 0.99
    assert np.isnan(tr2.estimate_pitch(x[None, :4096]))
    tr2.conf = 0.5
    assert abs(tr2.estimate_pitch(x[None, :4096]) - g["N4096_jump_f0"][0]) <= TOL_F0 * 200
    kat = pt.PitchTracker(RingBuffer(), fft_size=32, overlap=0.5)
    assert np.isnan(kat.estimate_pitch(g["kat32_frame"][None, :]))   # what the reference returns today (see the oracle)
    # dual-channel frames (the shared ring buffer with a second input): spectrum from row 0, gate level from BOTH rows
    # (pitch_tracker.py:404-407).  A tone 58 dB below full scale is unvoiced on its own and voiced beside a loud row.
    n = 4096 + 3 * 1024
    t = np.arange(n)
    quiet = 0.0018 * (np.sin(2 * np.pi * 220.0 * t / 48000.0) + 0.5 * np.sin(2 * np.pi * 440.0 * t / 48000.0))
    loud = 0.5 * np.random.default_rng(1).standard_normal(n)
    tr3 = pt.PitchTracker(RingBuffer())
    assert np.isnan(tr3.estimate_pitch(quiet[None, :4096]))
    f2 = tr3.estimate_pitch(np.stack([quiet, loud])[:, :4096])
    want2 = dsp.pitch_candidate(quiet[:4096], dsp.hann_symmetric(4096), tr3.logSpacedFreqs, tr3.kernels)[0]
    assert abs(f2 - want2) <= TOL_F0 * want2 and abs(f2 - 220.0) < 2.0
    buf2 = RingBuffer()
    tr4 = pt.PitchTracker(buf2)
    buf2.push(np.stack([quiet, loud]))
    assert tr4.update()
    got2 = tr4.get_estimates(3 * 1024 / 48000.0)
    ref = dsp.PitchGate()
    for f in range(4):
        seg = np.stack([quiet, loud])[:, f * 1024:f * 1024 + 4096]
        f0, c, _ = dsp.pitch_candidate(seg[0], dsp.hann_symmetric(4096), tr4.logSpacedFreqs, tr4.kernels)
        db = 20 * np.log10(np.sqrt(np.mean(seg ** 2)) + np.finfo(np.float64).eps)
        w = ref.step(f0, c, db)
        assert (np.isnan(w) and np.isnan(got2[f])) or abs(got2[f] - w) <= TOL_F0 * w, f


def test_other_grids_and_both_log_grid_kernels(pt, option):
    """The widget's grid (1023 points) runs the register-resident log-grid kernel; any other grid, and
    frt_set_option("pitch_grid_two_pass", 1), the two-pass one.  Both are the same arithmetic: identical bits on the widget's
    grid, oracle parity on a coarser and a finer grid (pitch_tracker.py:334-355 with other min_freq / cres)."""
    n_fft, hop, frames = 2048, 512, 48
    n = n_fft + hop * (frames - 1)
    t = np.arange(n)
    phase = 2 * np.pi * np.cumsum(180.0 * 2 ** (1.0 * t / n)) / 48000.0
    x = 0.2 * (np.sin(phase) + 0.5 * np.sin(2 * phase) + 0.25 * np.sin(3 * phase)) + 1e-3 * np.random.default_rng(5).standard_normal(n)
    fast, fast_raw = pt.PitchEngine(n_fft, hop).track(x, with_raw=True)
    option("pitch_grid_two_pass", 1)
    slow, slow_raw = pt.PitchEngine(n_fft, hop).track(x, with_raw=True)
    option("pitch_grid_two_pass", -1)
    assert np.array_equal(fast, slow, equal_nan=True) and np.array_equal(fast_raw, slow_raw, equal_nan=True)
    window = dsp.hann_symmetric(n_fft)
    for min_freq, max_freq, cres in [(100, 800, 20), (65, 1047, 7)]:
        freqs, kernels = dsp.swipe_tables(min_freq=min_freq, max_freq=max_freq, cres=cres)
        assert len(freqs) != 1023
        grid, _, kern = pt.swipe_tables(min_freq=min_freq, max_freq=max_freq, cres=cres)
        assert np.array_equal(grid, freqs) and np.array_equal(kern, kernels)
        f0, raw = pt.PitchEngine(n_fft, hop, grid=grid, kernels=kern).track(x, with_raw=True)
        for f in range(frames):
            r0, c, db = dsp.pitch_candidate(x[f * hop:f * hop + n_fft], window, freqs, kernels)
            assert abs(raw[0, 0, f] - r0) <= TOL_F0 * r0 and abs(raw[1, 0, f] - c) <= TOL_CONF and abs(raw[2, 0, f] - db) <= TOL_DB, (cres, f)
        gate = dsp.PitchGate()
        want = np.array([gate.step(raw[0, 0, f], raw[1, 0, f], raw[2, 0, f]) for f in range(frames)])
        assert np.array_equal(want, f0[0], equal_nan=True)
