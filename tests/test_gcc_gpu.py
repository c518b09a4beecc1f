"""K5 / G1-G3 parity: GCC-PHAT and the delay read-out against reference golden vectors and the oracle.

float64 on both sides with a different FFT factorisation: max|xcorr - ref| / max|ref| <= 1e-9 (the
north_star bar is 1e-5), identical argmax."""
import numpy as np
import pytest

from conftest import rel_max
from oracle import dsp

pytestmark = pytest.mark.gpu


def f64(x):
    return np.asarray(x, np.float32).astype(np.float64)


@pytest.mark.parametrize("L", [2400, 24000])
def test_gcc_against_golden(golden, hip, L):
    from friture_amd.signal.correlation import generalized_cross_correlation
    g = golden("gcc")
    d0, d1 = f64(g[f"L{L}_d0"]), f64(g[f"L{L}_d1"])
    keep0, keep1 = d0.copy(), d1.copy()
    x = generalized_cross_correlation(d0, d1)
    assert x.shape == (L,) and int(np.argmax(np.abs(x))) == int(g[f"L{L}_argmax"]) == 37
    if L == 2400:
        assert rel_max(x, g["L2400_xcorr"]) <= 1e-9
    else:
        scale = g["L24000_xcorr_norms"][0]
        assert np.max(np.abs(x[:128] - g["L24000_xcorr_head"])) <= 1e-9 * scale
        got = np.array([np.max(np.abs(x)), np.sqrt(np.sum(x ** 2)), np.std(x)])
        assert np.max(np.abs(got / g["L24000_xcorr_norms"] - 1)) < 1e-9
    # the in-place mean removal of the reference (correlation.py:27-28)
    assert np.allclose(d0, keep0 - keep0.mean(), rtol=0, atol=1e-15)
    assert np.allclose(d1, keep1 - keep1.mean(), rtol=0, atol=1e-15)


@pytest.mark.parametrize("L", [4, 6, 30, 1024, 540, 12288, 16384, 49152])
def test_gcc_lengths_against_oracle(hip, L):
    """radix mix, the R = 1 / 2 / 4 splits, tiny windows"""
    from friture_amd.signal.correlation import GccPhat
    rng = np.random.default_rng(L)
    d0 = 0.25 * rng.standard_normal((3, L))
    lag = min(5, L // 4)
    d1 = np.roll(d0, lag, axis=1) + 0.02 * rng.standard_normal((3, L))
    d1[2] = -d1[2]                                  # inverted polarity: negative peak
    x, am = GccPhat(L, 3).correlate(d0, d1)
    for p in range(3):
        ref, _, _ = dsp.gcc_phat(d0[p], d1[p])
        assert rel_max(x[p], ref) <= 1e-9, (L, p, rel_max(x[p], ref))
        assert am[p] == int(np.argmax(np.abs(ref)))
    if L >= 30:
        assert list(am) == [lag] * 3 and x[2, lag] < 0 < x[0, lag]


@pytest.mark.parametrize("L", [540, 24000, 49152])
@pytest.mark.parametrize("one_workgroup", ["0", "1"])
def test_gcc_signals_with_a_large_mean(hip, L, one_workgroup, option):
    """The one-workgroup kernel takes the means while the samples pass into the sub-transforms and removes them in the spectrum
    (mean x rfft(window), csrc/gcc.hip); the launch shape with sub-transforms as workgroups of their own subtracts them in time.
    Signals whose mean is 10-20 times their deviation (the leakage of the mean's window spectrum reaches every bin), both shapes,
    R = 1 / 2 / 4: same tolerance, same means as numpy."""
    from friture_amd.signal.correlation import GccPhat
    option("gcc_one_workgroup", int(one_workgroup))
    rng = np.random.default_rng(7 * L)
    d0 = 0.25 * rng.standard_normal((2, L)) + 2.5
    d1 = np.roll(d0, 9, axis=1) - 2.5 - 4.0 + 0.02 * rng.standard_normal((2, L))
    x, am = GccPhat(L, 2).correlate(d0, d1)
    for p in range(2):
        ref, _, _ = dsp.gcc_phat(d0[p], d1[p])
        assert rel_max(x[p], ref) <= 1e-9, (L, p, rel_max(x[p], ref))
        assert am[p] == int(np.argmax(np.abs(ref))) == 9


@pytest.mark.parametrize("L", [2640, 14336, 26400, 50400, 232800, 1_200_000])
def test_gcc_any_window_length(hip, L):
    """The delay-range spin box gives windows of 2400 r samples, r = 1..10000 (delay_estimator.py:114-115,222-226): most
    are not 5-smooth (0.11 s -> 2640 = 2^4 3 5 11; 9.7 s -> 232800 = 2400 x 97) or too long for the one-workgroup kernel
    (2.1 s -> 50400; 50 s -> 1.2 M).  numpy takes every length; those go through the chirp-z path: same tolerance."""
    from friture_amd.signal.correlation import GccPhat
    rng = np.random.default_rng(L)
    P = 2 if L < 500000 else 1
    d0 = 0.25 * rng.standard_normal((P, L))
    d1 = np.roll(d0, 41, axis=1) + 0.02 * rng.standard_normal((P, L))
    if P == 2:
        d1[1] = -d1[1]
    x, am = GccPhat(L, P).correlate(d0, d1)
    for p in range(P):
        ref, _, _ = dsp.gcc_phat(d0[p], d1[p])
        assert rel_max(x[p], ref) <= 1e-9, (L, p, rel_max(x[p], ref))
        assert am[p] == int(np.argmax(np.abs(ref))) == 41


def test_gcc_chirp_path_equals_one_workgroup_path(hip, option):
    """L = 24000 (the widget's default window) through both implementations."""
    from friture_amd.signal.correlation import GccPhat
    rng = np.random.default_rng(5)
    d0 = 0.25 * rng.standard_normal((2, 24000))
    d1 = np.roll(d0, 37, axis=1) + 0.05 * rng.standard_normal((2, 24000))
    x_fast, am_fast = GccPhat(24000, 2).correlate(d0, d1)
    option("gcc_any_length", 1)
    x_any, am_any = GccPhat(24000, 2).correlate(d0, d1)
    assert list(am_fast) == list(am_any) == [37, 37]
    assert rel_max(x_any, x_fast) <= 1e-10


def test_delay_estimator_other_ranges(hip):
    """Delay_Estimator mirror with a 0.3 s range (L = 7200) and a 1.1 s range (L = 26400, chirp-z path)."""
    from friture_amd.delay_estimator import DelayEstimator
    for rng_s in (0.3, 1.1):
        de = DelayEstimator()
        de.set_delayrange(rng_s)
        rng = np.random.default_rng(int(rng_s * 10))
        n = int(48000 * rng_s * 4.5)
        a = (0.25 * rng.standard_normal(n)).astype(np.float32)
        b = np.roll(a, 4 * 25) + (0.01 * rng.standard_normal(n)).astype(np.float32)     # 25 samples at the decimated rate
        for pos in range(0, n - 512, 512):
            de.handle_new_data(np.stack([a[pos:pos + 512], b[pos:pos + 512]]).astype(np.float64))
        assert abs(de.delay_ms - 1e3 * 25 / 12000.0) < 0.2, (rng_s, de.delay_ms)


def test_gcc_errors(hip):
    from friture_amd._lib import FritureHipError
    from friture_amd.signal.correlation import GccPhat, generalized_cross_correlation
    with pytest.raises(FritureHipError):
        GccPhat(1001, 1)                 # odd length
    with pytest.raises(ValueError):
        generalized_cross_correlation(np.zeros(100), np.zeros(98))


def test_readout_against_oracle(hip):
    from friture_amd.signal.correlation import GccPhat
    L, rate, rng_s = 24000, 12000.0, 1.0
    rng = np.random.default_rng(1)
    g = GccPhat(L, 2)
    d0 = 0.25 * rng.standard_normal((2, L))
    d1 = np.stack([np.roll(d0[0], 37), -np.roll(d0[1], L - 50)]) + 0.05 * rng.standard_normal((2, L))
    x, _ = g.correlate(d0, d1)
    old = None
    for it in range(3):
        sm, ro = g.readout(x, old, rate, rng_s)
        for p in range(2):
            ref = dsp.delay_readout(x[p], None if old is None else old[p], rate, rng_s)
            assert np.max(np.abs(sm[p] - ref["smoothed"])) <= 1e-15
            assert ro[p].argmax == ref["argmax"] and ro[p].correlation_pct == ref["correlation_pct"]
            assert abs(ro[p].delay_ms - ref["delay_ms"]) < 1e-9 and abs(ro[p].extremum - ref["extremum"]) < 1e-12
            assert abs(ro[p].distance_m - ref["distance_m"]) < 1e-9
        old = sm
        x = x * 0.9
    assert abs(ro[0].delay_ms - 1e3 * 37 / rate) < 1e-9 and ro[0].extremum > 0
    assert abs(ro[1].delay_ms + 1e3 * 50 / rate) < 1e-9 and ro[1].extremum < 0      # wrapped negative delay, inverted


def test_delay_estimator_chain_against_oracle(hip):
    """The whole widget body on 2-channel 48 kHz chunks vs the same chain built from the oracle."""
    from friture_amd.delay_estimator import DelayEstimator
    rng = np.random.default_rng(3)
    n = 512 * 60
    a = (0.25 * rng.standard_normal(n)).astype(np.float32).astype(np.float64)
    b = np.roll(a, 4 * 21) + 0.01 * rng.standard_normal(n)        # 21 samples at 12 kHz
    est = DelayEstimator(delayrange_s=0.25)                       # window 6000, hop 3000 decimated samples
    t = dsp.load_filter_tables()
    z0, z1 = dsp.decimate_multiple_filtic(2, t["bdec"], t["adec"]), dsp.decimate_multiple_filtic(2, t["bdec"], t["adec"])
    r0, r1 = dsp.MirrorRing(), dsp.MirrorRing()
    old, old_index, ref = None, 0, None
    for c in range(60):
        chunk = np.stack([a[c * 512:(c + 1) * 512], b[c * 512:(c + 1) * 512]])
        est.handle_new_data(chunk)
        y0, z0 = dsp.decimate_multiple(2, t["bdec"], t["adec"], chunk[0], z0)
        y1, z1 = dsp.decimate_multiple(2, t["bdec"], t["adec"], chunk[1], z1)
        r0.push(y0[None, :])
        r1.push(y1[None, :])
        while r0.offset - old_index >= 3000:
            old_index += 3000
            w0, w1 = r0.data_indexed(old_index, 6000)[0], r1.data_indexed(old_index, 6000)[0]
            xc, m0, m1 = dsp.gcc_phat(w0, w1)
            w0[:], w1[:] = m0, m1                                   # the reference de-means the ring views in place
            ref = dsp.delay_readout(xc, old, 12000.0, 0.25)
            old = ref["smoothed"]
    assert ref is not None and abs(est.delay_ms - ref["delay_ms"]) < 1e-9 and abs(est.delay_ms - 1.75) < 1e-9
    assert est.correlation == ref["correlation_pct"] and abs(est.Xcorr_extremum - ref["extremum"]) < 1e-9
    assert np.max(np.abs(est.old_Xcorr - ref["smoothed"])) < 1e-9


def test_configs4_full_size_properties(hip):
    """BASELINE configs[4], the delay-estimator half: 100 windows of L = 24000 at 50 % overlap of one 2-channel
    recording whose second channel is the first delayed by 37 samples, device resident.
    Properties: every window finds the true delay, swapping the channels negates it (circularly), scaling both
    channels leaves the PHAT-weighted correlation unchanged to rounding, sampled windows equal the oracle."""
    import torch
    from friture_amd.signal.correlation import GccPhat
    L, windows, lag = 24000, 100, 37
    n = L // 2 * (windows + 1)
    rng = np.random.default_rng(123)
    ch0 = 0.25 * rng.standard_normal(n)
    ch1 = np.roll(ch0, lag) + 0.025 * rng.standard_normal(n)
    d0 = np.stack([ch0[w * L // 2:w * L // 2 + L] for w in range(windows)])
    d1 = np.stack([ch1[w * L // 2:w * L // 2 + L] for w in range(windows)])
    g = GccPhat(L, windows)
    a0, a1 = torch.from_numpy(d0).cuda(), torch.from_numpy(d1).cuda()
    x01, am01 = g.correlate(a0, a1)
    x10, am10 = g.correlate(a1, a0)
    xs, _ = g.correlate(4.0 * a0, 4.0 * a1)
    torch.cuda.synchronize()
    assert bool((am01 == lag).all()) and bool((am10 == L - lag).all())
    x01n = x01.cpu().numpy()
    assert np.max(np.abs(xs.cpu().numpy() - x01n)) <= 1e-12 * np.max(np.abs(x01n))
    for w in (0, 41, 99):
        ref, _, _ = dsp.gcc_phat(d0[w].copy(), d1[w].copy())
        assert np.max(np.abs(x01n[w] - ref)) <= 1e-9 * np.max(np.abs(ref))


def test_gcc_large_batch_one_workgroup_per_pair(hip):
    """300 window pairs of L = 24000: above 160 pairs the default dispatch is the one-workgroup kernel (compile-time
    6000-point plan, cross spectrum in registers, last sub-spectrum and packed inverse input in LDS).  Every window finds the
    delay, the means it reports are the windows' means, sampled windows (first, one inside, last — the last workgroup of the
    grid) equal the oracle to 1e-9, a window with a large DC offset included, and unaligned windows (views at an odd sample
    offset, as a ring hands them out) give the same correlation as their aligned copies."""
    import torch
    from friture_amd.signal.correlation import GccPhat
    L, pairs, lag = 24000, 300, 23
    rng = np.random.default_rng(77)
    d0 = 0.25 * rng.standard_normal((pairs, L))
    d1 = np.roll(d0, lag, axis=1) + 0.02 * rng.standard_normal((pairs, L))
    d0[150] += 3.0                                              # DC offsets: the means matter
    d1[150] -= 1.5
    g = GccPhat(L, pairs)
    a0, a1 = torch.from_numpy(d0).cuda(), torch.from_numpy(d1).cuda()
    x, am = g.correlate(a0, a1)
    torch.cuda.synchronize()
    assert bool((am == lag).all())
    means = g.means.cpu().numpy()
    assert np.max(np.abs(means[:, 0] - d0.mean(axis=1))) <= 1e-14 and np.max(np.abs(means[:, 1] - d1.mean(axis=1))) <= 1e-14
    xn = x.cpu().numpy()
    for w in (0, 150, 299):
        ref, _, _ = dsp.gcc_phat(d0[w].copy(), d1[w].copy())
        assert np.max(np.abs(xn[w] - ref)) <= 1e-9 * np.max(np.abs(ref)), w
    # the same windows at an 8-byte (not 16-byte) aligned address: one pair, views into a longer buffer
    g1 = GccPhat(L, 1)
    buf0 = torch.zeros(L + 1, dtype=torch.float64, device="cuda")
    buf1 = torch.zeros(L + 1, dtype=torch.float64, device="cuda")
    buf0[1:] = a0[7]
    buf1[1:] = a1[7]
    xu, amu = g1.correlate(buf0[1:][None, :], buf1[1:][None, :])
    xa, ama = g1.correlate(a0[7:8].clone(), a1[7:8].clone())
    torch.cuda.synchronize()
    assert int(amu[0]) == int(ama[0]) == lag
    assert np.max(np.abs(xu.cpu().numpy() - xa.cpu().numpy())) <= 1e-12 * np.max(np.abs(xn[7]))


def test_delay_estimator_stream_equals_host_chain(hip):
    """DelayEstimatorStream (decimator states, 12 kHz rings, windows, correlation and read-out resident in HBM) against
    DelayEstimator (host rings, every stage a host-staged call): the same read-out chunk after chunk — including the ring
    growth of the default 1 s range (24000-sample windows in a 10000-sample ring) and the in-place mean removal."""
    from friture_amd.delay_estimator import DelayEstimator, DelayEstimatorStream
    for rng_s in (0.3, 1.0):
        a, b = DelayEstimator(rng_s), DelayEstimatorStream(rng_s)
        rng = np.random.default_rng(int(rng_s * 100))
        n = int(48000 * rng_s * 5.2)
        x0 = (0.25 * rng.standard_normal(n) + 0.01).astype(np.float32)        # a DC offset: the means matter
        x1 = np.roll(x0, 4 * 31) - 0.02 + (0.01 * rng.standard_normal(n)).astype(np.float32)
        windows = 0
        for pos in range(0, n - 512, 512):
            chunk = np.stack([x0[pos:pos + 512], x1[pos:pos + 512]]).astype(np.float64)
            a.handle_new_data(chunk)
            b.handle_new_data(chunk)
            assert a.correlation == b.correlation and abs(a.delay_ms - b.delay_ms) <= 1e-12, (rng_s, pos)
            assert abs(a.Xcorr_extremum - b.Xcorr_extremum) <= 1e-12 * max(abs(a.Xcorr_extremum), 1e-30)
            windows += a.delay_ms != 0.
        assert windows > 0 and abs(a.delay_ms - 1e3 * 31 / 12000.0) < 0.1


def test_delay_object_rings_equal_host_rings(hip):
    """frt_delay_* (decimation states and both 12 kHz mirror rings inside one C object, pushes asynchronous): the windows it
    hands out hold the same bits as the host chain's RingBuffer views — before the rings have filled (zeros), across the
    growth from 10000 samples, with ragged chunk lengths, and after the in-place mean removal of earlier windows."""
    from friture_amd.delay_estimator import DelayEstimator, DelayEstimatorStream
    a, b = DelayEstimator(0.4), DelayEstimatorStream(0.4)
    rng = np.random.default_rng(5)
    pos_total = 0
    for i, n in enumerate([512, 512, 37, 1, 1024, 4096, 511] * 12):
        x0 = 0.3 * rng.standard_normal(n) + 0.05
        chunk = np.stack([x0, 0.5 * x0 + 0.01 * rng.standard_normal(n)])
        a.handle_new_data(chunk)
        b.handle_new_data(chunk)
        assert a.ringbuffer0.offset == b.offset
        if i % 5 == 0:
            for end, length in ((b.offset, 300), (b.offset - b.offset // 3, 2000), (b.offset, 9000 + 700 * (i // 5))):
                ref = np.concatenate([a.ringbuffer0.data_indexed(end, length), a.ringbuffer1.data_indexed(end, length)])
                assert np.array_equal(b.window(end, length), ref), (i, end, length)
    assert a.delay_ms == b.delay_ms and a.correlation == b.correlation and a.correlation > 0
    # the stream object's rings stay inspectable like the reference widget's (ringbuffer0 / ringbuffer1: offset, data_indexed)
    assert b.ringbuffer0.offset == a.ringbuffer0.offset and b.ringbuffer1.offset == a.ringbuffer1.offset
    for end, length in ((b.offset, 128), (b.offset - 1000, 2500)):
        assert np.array_equal(b.ringbuffer0.data_indexed(end, length), a.ringbuffer0.data_indexed(end, length))
        assert np.array_equal(b.ringbuffer1.data_indexed(end, length), a.ringbuffer1.data_indexed(end, length))


def test_delay_object_gate_on_constant_windows(hip):
    """The std of the gate of delay_estimator.py:127 as the device object computes it (frt_delay_window_std): numpy's value
    on ordinary windows; exactly 0 on all-zero windows (the reference's silent case) AND on constant non-zero windows — for
    which numpy's own std is 0 or rounding noise depending on the value and the summation order."""
    import ctypes

    import torch

    from friture_amd import _lib
    from friture_amd.delay_estimator import DelayEstimatorStream
    b = DelayEstimatorStream(0.1)
    b.handle_new_data(np.zeros((2, 512)))
    rng = np.random.default_rng(17)
    L = 4800
    noise = rng.standard_normal(L)
    one_off = np.full(L, 0.3)
    one_off[L - 7] = 0.3000001
    for w0, w1 in ((np.zeros(L), np.full(L, 0.1)), (np.full(L, 0.5), noise), (one_off, np.full(L, -3.7e-9))):
        t0, t1 = torch.from_numpy(w0).cuda(), torch.from_numpy(w1).cuda()
        stds = (ctypes.c_double * 2)()
        _lib.check(hip.frt_delay_window_std(b._h, ctypes.c_void_p(t0.data_ptr()), ctypes.c_void_p(t1.data_ptr()), L, stds))
        for got, w in zip(stds, (w0, w1)):
            if np.all(w == w[0]):
                assert got == 0.0
            else:
                assert got > 0.0 and abs(got - np.std(w)) <= 1e-12 * np.std(w)
    # and the silent chain end to end: nothing is estimated
    for _ in range(24):
        b.handle_new_data(np.zeros((2, 512)))
    assert b.delay_ms == 0. and b.correlation == 0 and b.Xcorr_extremum == 0.


def test_delay_object_argument_checks(hip):
    import ctypes

    from friture_amd import _lib
    from friture_amd.delay_estimator import DelayEstimatorStream
    b = DelayEstimatorStream(0.4)
    p0, p1 = ctypes.c_void_p(), ctypes.c_void_p()
    with pytest.raises(_lib.FritureHipError):
        _lib.check(hip.frt_delay_window(b._h, 10, 100, ctypes.byref(p0), ctypes.byref(p1)))      # nothing pushed yet
    b.handle_new_data(np.zeros((2, 512)))
    assert b.offset == 128
    b.handle_new_data(np.zeros((1, 512)))            # mono: no push (delay_estimator.py:89-91)
    assert b.offset == 128 and not b.two_channels
    b.handle_new_data(np.zeros((2, 0)))
    assert b.offset == 128


def test_gcc_small_batch_path_equals_one_workgroup_path(hip, option):
    """Batches that would leave most CUs idle run a pair's phases as launches of their own (forward per signal, cross
    spectrum, packing, inverse per sub-transform); the one-workgroup kernel keeps the cross spectrum in registers and the
    last sub-spectrum / the packed inverse input in LDS.  Same arithmetic up to the order of the means' block sums and the
    compiler's contraction choices: 1e-12 of the correlation's scale, identical arg-max, for R = 1, 2 and 4."""
    from friture_amd.signal.correlation import GccPhat
    for L in (2400, 24000, 49152):
        rng = np.random.default_rng(L)
        d0 = 0.25 * rng.standard_normal((3, L))
        d1 = np.roll(d0, 17, axis=1) + 0.05 * rng.standard_normal((3, L))
        option("gcc_one_workgroup", 0)
        x_multi, am_multi = GccPhat(L, 3).correlate(d0, d1)
        option("gcc_one_workgroup", 1)
        x_one, am_one = GccPhat(L, 3).correlate(d0, d1)
        option("gcc_one_workgroup", -1)
        ref, _, _ = dsp.gcc_phat(d0[1].copy(), d1[1].copy())
        for x in (x_multi, x_one):
            assert np.max(np.abs(x[1] - ref)) <= 1e-9 * np.max(np.abs(ref)), L
        assert np.max(np.abs(x_multi - x_one)) <= 1e-12 * np.max(np.abs(x_one)), L
        assert list(am_multi) == list(am_one) == [17] * 3


def test_gcc_default_window_four_way_small_batches_equal_the_two_way_split(hip):
    """Handles of at most CUs / 8 pairs of the default window (24000 samples) run the launches-of-phases path with FOUR
    sub-transforms of 3000 points (compile-time plan 3 x 10 x 10 x 10; the means from the forward workgroups' partial sums, removed
    in the spectrum), larger ones with two of 6000: the same pairs through a 32-pair handle and through the first 32 pairs of a
    33-pair handle, with large means — 1e-13 of the correlation (measured 4e-16)'s scale, identical arg-max and means, and the oracle on one pair."""
    from friture_amd.signal.correlation import GccPhat
    L = 24000
    rng = np.random.default_rng(2405)
    d0 = 0.25 * rng.standard_normal((33, L)) + rng.uniform(-3.0, 3.0, (33, 1))
    d1 = np.roll(d0, 29, axis=1) + 0.05 * rng.standard_normal((33, L)) - 1.5
    x4, am4 = GccPhat(L, 32).correlate(d0[:32].copy(), d1[:32].copy())
    x2, am2 = GccPhat(L, 33).correlate(d0.copy(), d1.copy())
    assert np.max(np.abs(x4 - x2[:32])) <= 1e-13 * np.max(np.abs(x2)), np.max(np.abs(x4 - x2[:32])) / np.max(np.abs(x2))
    assert list(am4) == list(am2[:32]) == [29] * 32
    ref, _, _ = dsp.gcc_phat(d0[7].copy(), d1[7].copy())
    assert np.max(np.abs(x4[7] - ref)) <= 1e-9 * np.max(np.abs(ref))
    one, am1 = GccPhat(L, 1).correlate(d0[7:8].copy(), d1[7:8].copy())
    assert np.max(np.abs(one[0] - ref)) <= 1e-9 * np.max(np.abs(ref)) and int(am1[0]) == 29


def test_gcc_four_way_handle_forced_onto_the_one_workgroup_kernel(hip, option):
    """A small-batch handle of the default window is made with four sub-transforms of 3000 points; forcing the one-workgroup kernel on it
    AFTER it was made runs that kernel's R = 4 instance on the run-time plan — slower, and the same correlation."""
    from friture_amd.signal.correlation import GccPhat
    L = 24000
    rng = np.random.default_rng(77)
    d0 = 0.25 * rng.standard_normal((2, L)) + 0.5
    d1 = np.roll(d0, 11, axis=1) + 0.03 * rng.standard_normal((2, L))
    g = GccPhat(L, 2)
    x_phases, am_phases = g.correlate(d0.copy(), d1.copy())
    option("gcc_one_workgroup", 1)
    x_one, am_one = g.correlate(d0.copy(), d1.copy())
    option("gcc_one_workgroup", -1)
    assert np.max(np.abs(x_phases - x_one)) <= 1e-12 * np.max(np.abs(x_one))
    assert list(am_phases) == list(am_one) == [11, 11]
    ref, _, _ = dsp.gcc_phat(d0[0].copy(), d1[0].copy())
    assert np.max(np.abs(x_one[0] - ref)) <= 1e-9 * np.max(np.abs(ref))


def test_gcc_resident_kernel_equals_the_slab_kernel_and_the_oracle(hip, option):
    """The default window with one workgroup per pair runs gcc_phat_resident_kernel (csrc/gcc_resident.h): nothing passes through HBM
    between the signals and the correlation — bins owned by quads, signal 0's sub-spectra parked in registers, signal 1's in LDS.
    Against gcc_phat_kernel (option gcc_resident = 0: the sub-spectra through a scratch slab) on the same pairs — large and opposite
    means, an inverted pair, a pair of impulses (a flat cross spectrum) — 1e-12 of the correlation's scale, identical arg-max and means; and
    the oracle on every pair."""
    from friture_amd.signal.correlation import GccPhat
    L, P = 24000, 5
    rng = np.random.default_rng(606)
    d0 = 0.25 * rng.standard_normal((P, L)) + rng.uniform(-3.0, 3.0, (P, 1))
    d1 = np.roll(d0, 37, axis=1) + 0.05 * rng.standard_normal((P, L)) - 1.5
    d1[2] = -d1[2]
    d0[4] = 0.0
    d0[4, 5000] = 1.0
    d1[4] = 0.0
    d1[4, 5063] = 0.5
    option("gcc_one_workgroup", 1)
    g = GccPhat(L, P)
    x_res, am_res = g.correlate(d0.copy(), d1.copy())
    option("gcc_resident", 0)
    x_slab, am_slab = g.correlate(d0.copy(), d1.copy())
    option("gcc_resident", -1)
    scale = np.max(np.abs(x_slab))
    assert np.max(np.abs(x_res - x_slab)) <= 1e-12 * scale, np.max(np.abs(x_res - x_slab)) / scale
    assert list(am_res) == list(am_slab)
    for p in range(P):
        ref, _, _ = dsp.gcc_phat(d0[p].copy(), d1[p].copy())
        assert np.max(np.abs(x_res[p] - ref)) <= 1e-9 * np.max(np.abs(ref)), p
        assert int(am_res[p]) == int(np.argmax(np.abs(ref)))
    assert list(am_res[:4]) == [37] * 4 and int(am_res[4]) == 63 and x_res[2, 37] < 0 < x_res[0, 37]


def test_gcc_resident_kernel_on_windows_that_are_views(hip, option):
    """Device windows that start 8 bytes off a 16-byte boundary (views into a ring): the resident kernel's 16-byte buffer accesses
    need 4-byte alignment only — same correlation as the aligned copy, bit for bit."""
    import torch
    from friture_amd.signal.correlation import GccPhat
    L, P = 24000, 3
    rng = np.random.default_rng(11)
    d0 = 0.25 * rng.standard_normal((P, L))
    d1 = np.roll(d0, 21, axis=1) + 0.05 * rng.standard_normal((P, L))
    option("gcc_one_workgroup", 1)
    g = GccPhat(L, P)
    dev = torch.device("cuda", 0)
    a0, a1 = torch.from_numpy(d0).to(dev), torch.from_numpy(d1).to(dev)
    x_al, am_al = g.correlate(a0, a1)
    x_al = x_al.clone()
    pad0, pad1 = torch.zeros(P * L + 1, dtype=torch.float64, device=dev), torch.zeros(P * L + 1, dtype=torch.float64, device=dev)
    pad0[1:] = a0.reshape(-1)
    pad1[1:] = a1.reshape(-1)
    v0, v1 = pad0[1:].view(P, L), pad1[1:].view(P, L)
    assert v0.data_ptr() % 16 == 8
    x_un, am_un = g.correlate(v0, v1)
    assert torch.equal(x_un, x_al) and list(am_un.cpu().numpy()) == list(am_al.cpu().numpy()) == [21] * P
