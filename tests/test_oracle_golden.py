"""The oracle against golden vectors recorded from the unmodified reference (oracle/make_golden.py).

These run wherever the tests run — in particular on the GPU box, where the reference checkout does
not exist — and pin oracle/dsp.py to the reference's outputs for every row of SURVEY.md §8a.
"""
import numpy as np
import pytest

from oracle import dsp


def f64(x):
    return np.asarray(x, np.float32).astype(np.float64)


@pytest.mark.parametrize("key,n_fft,hop", [("N32_hop16_noise", 32, 16), ("N256_hop64_tone", 256, 64),
                                           ("N1024_hop512_noise", 1024, 512), ("N1024_hop256_tone", 1024, 256),
                                           ("N4096_hop1024_noise", 4096, 1024), ("N16384_hop8192_tone", 16384, 8192)])
def test_psd(golden, key, n_fft, hop):
    g = golden("psd")
    assert np.array_equal(dsp.stft_psd(f64(g[key + "_x"]), n_fft, hop), g[key + "_psd"])


def test_weighting_and_axes(golden):
    g = golden("psd")
    f = dsp.frequency_axis(1024)
    assert np.array_equal(f, g["N1024_freq"])
    for got, want in zip(dsp.weighting_curves(f), (g["N1024_A"], g["N1024_B"], g["N1024_C"])):
        assert np.array_equal(got, want)


def test_image(golden):
    g = golden("image")
    lut = dsp.colour_lut(dsp.cmrmap())
    assert np.array_equal(lut, g["lut"])
    img = dsp.spectrogram_image(f64(g["x"]), 1024, 512, g["weight"], float(g["spec_min"]), float(g["spec_max"]), lut)
    assert np.array_equal(img.T, g["image"])


def test_resamplers(golden):
    g = golden("pipeline")
    for scale in ("linear", "log", "mel", "erb", "octave"):
        tg = dsp.frequency_targets(scale, 20.0, 20000.0, 100)
        assert np.array_equal(tg, g[f"fr_{scale}_targets"])
        assert np.array_equal(dsp.frequency_resample(tg, g["freq"], g["norm"]), g[f"fr_{scale}"])
    for tag, (L, M) in {"down": (25, 16), "up": (3, 7)}.items():
        tr = dsp.TimeResampler(L, M, 100)
        assert np.array_equal(tr.push(g["fr_mel"][:, :5]), g[f"tr_{tag}_a"])
        assert np.array_equal(tr.push(g["fr_mel"][:, 5:]), g[f"tr_{tag}_b"])
    # a resize between pushes: the carried column is Fourier-resampled (scipy_resample.py:108-141)
    col = g["fr_mel"][:, 7]
    for h in (137, 64, 100, 211):
        assert np.max(np.abs(dsp.fourier_resample(col, h) - g[f"fourier_100_{h}"])) <= 1e-13
    tr = dsp.TimeResampler(25, 16, 100)
    for i in range(3):
        out = tr.push(g[f"tr_resize_in_{i}"])
        assert out.shape == g[f"tr_resize_out_{i}"].shape
        assert np.max(np.abs(out - g[f"tr_resize_out_{i}"])) <= 1e-13


def test_exp_smoothing(golden):
    g = golden("exp_smoothing")
    assert dsp.exp_smoothed_value(g["kern"], 0.02, g["d1"], 0.3) == float(g["r1"])
    assert np.array_equal(dsp.exp_smoothed_value_2d(g["kern"], 0.02, g["d2"], g["prev"]), g["r2"])
    assert np.array_equal(dsp.exp_smoothed_value_2d(g["kern"], 0.02, g["d2"][:, :17], g["prev"]), g["r3"])


def test_decimate_multiple(golden):
    g = golden("iir")
    t = dsp.load_filter_tables()
    x = f64(g["x_dec"])
    for force in (False, True):
        zs = dsp.decimate_multiple_filtic(2, t["bdec"], t["adec"])
        for c in range(4):
            if force:   # pure-Python loop, chunk 0 only (slow)
                y, _ = dsp.lfilter_df2t(t["bdec"], t["adec"], x[:512], np.zeros(12), force_python=True)
                y2, _ = dsp.lfilter_df2t(t["bdec"], t["adec"], y[::2], np.zeros(12), force_python=True)
                assert np.array_equal(y2[::2], g["dec2_0"])
                break
            y, zs = dsp.decimate_multiple(2, t["bdec"], t["adec"], x[c * 512:(c + 1) * 512], zs)
            assert np.array_equal(y, g[f"dec2_{c}"])


@pytest.mark.parametrize("bpo", [1, 3, 6, 12, 24])
def test_iir_bank(golden, bpo):
    g = golden("iir")
    t = dsp.load_filter_tables()
    boct, aoct = list(t[f"boct_{bpo}"]), list(t[f"aoct_{bpo}"])
    zs = dsp.iir_bank_filtic(t["bdec"], t["adec"], boct, aoct)
    x = f64(g[f"bank{bpo}_x"])
    for blk in range(2):
        y, dec, zs = dsp.iir_bank(t["bdec"], t["adec"], boct, aoct, x[blk * 1024:(blk + 1) * 1024], zs)
        assert np.array_equal(np.array([np.sum(v ** 2) for v in y]), g[f"bank{bpo}_energy_{blk}"])
        if bpo == 3:
            for k in range(27):
                assert np.array_equal(y[k], g[f"bank3_y_{blk}_{k}"])
    assert np.array_equal(np.array(dec), g[f"bank{bpo}_dec"])
    assert np.array_equal(np.concatenate(zs), g[f"bank{bpo}_zf"])


@pytest.mark.parametrize("bpo", [1, 3, 6, 12, 24])
def test_ola_bank(golden, bpo):
    g = golden("ola")
    bank = dsp.OlaBank(bpo)
    x = f64(g[f"ola{bpo}_x"])
    pos = 0
    for blk, n in enumerate([1024, 512, 1024]):
        y, dec = bank.filter(x[pos:pos + n])
        pos += n
        e = np.array([np.sum(v ** 2) for v in y])
        # the oracle recomputes H = rfft(taps) instead of loading upstream's table: 1e-15 differences
        assert np.max(np.abs(e / g[f"ola{bpo}_energy_{blk}"] - 1)) < 1e-12
        if bpo == 3:
            for k in range(27):
                assert np.max(np.abs(y[k] - g[f"ola3_y_{blk}_{k}"])) <= 1e-12 * np.max(np.abs(g[f"ola3_y_{blk}_{k}"]))
    assert np.array_equal(np.array(dec), g[f"ola{bpo}_dec"])
    fi, flo, fhi = dsp.octave_frequencies(9 * bpo, bpo)
    assert np.array_equal(fi, g[f"bands{bpo}_fi"]) and np.array_equal(flo, g[f"bands{bpo}_flow"])
    assert np.array_equal(fhi, g[f"bands{bpo}_fhigh"])
    for got, want in zip(dsp.band_weighting(fi), (g[f"bands{bpo}_A"], g[f"bands{bpo}_B"], g[f"bands{bpo}_C"])):
        assert np.array_equal(got, want)


def test_gcc_phat(golden):
    g = golden("gcc")
    x, _, _ = dsp.gcc_phat(f64(g["L2400_d0"]), f64(g["L2400_d1"]))
    assert np.array_equal(x, g["L2400_xcorr"])
    assert int(np.argmax(np.abs(x))) == int(g["L2400_argmax"]) == 37
    x, _, _ = dsp.gcc_phat(f64(g["L24000_d0"]), f64(g["L24000_d1"]))
    assert np.array_equal(x[:128], g["L24000_xcorr_head"])
    assert np.array_equal(np.array([np.max(np.abs(x)), np.sqrt(np.sum(x ** 2)), np.std(x)]), g["L24000_xcorr_norms"])
    assert int(np.argmax(np.abs(x))) == int(g["L24000_argmax"]) == 37


def test_ring(golden):
    g = golden("ring")
    ring = dsp.MirrorRing()
    for step in range(6):
        ring.push(g[f"blk{step}"])
        ln = min(ring.offset, 4096)
        assert np.array_equal(ring.data_indexed(ring.offset - 100, ln - 100), g[f"win{step}"])
    with pytest.raises(Exception):
        dsp.decimate(np.ones(3), np.ones(3), np.zeros(0))


def test_spectrum_readout(golden):
    g = golden("spectrum")
    alpha = float(g["kern_alpha"])
    ro = dsp.spectrum_readout(g["spn"], dsp.smoothing_kernel(alpha, 8192), alpha, np.zeros(513), g["weight"],
                              dsp.frequency_axis(1024))
    assert np.array_equal(ro["smoothed"], g["smoothed"]) and np.array_equal(ro["db"], g["db"])
    assert np.array_equal(dsp.harmonic_product_spectrum(g["smoothed"]), g["hps"])
    assert ro["peak_index"] == int(g["peak_index"]) and ro["pitch_index"] == int(g["pitch_index"])


PITCH_CASES = [(4096, 1024), (2048, 1024), (1024, 512)]
PITCH_SIGNALS = ["steady220", "glide", "jump", "quiet", "noise", "silence", "high900"]


@pytest.mark.parametrize("n_fft,hop", PITCH_CASES)
def test_pitch_tracker(golden, n_fft, hop):
    """T1: table construction and per-frame estimates vs the reference's
    PitchTracker executed in the build container (oracle/make_golden_pitch.py)."""
    g = golden("pitch")
    freqs, kernels = dsp.swipe_tables()
    assert np.array_equal(freqs, g["freqs"])
    assert np.array_equal(kernels[g["kernel_rows"]], g["kernel_sample"])
    assert np.array_equal([kernels.sum(), np.abs(kernels).sum(), (kernels ** 2).sum()], g["kernel_sha_sum"])
    with np.errstate(invalid="ignore"):
        for name in PITCH_SIGNALS:
            key = f"N{n_fft}_{name}"
            out = dsp.pitch_track(g[key + "_x"], n_fft, hop, freqs, kernels)
            assert np.array_equal(out[0], g[key + "_f0"], equal_nan=True), key
            assert np.array_equal(out[1], g[key + "_raw"], equal_nan=True), key


def test_pitch_upstream_known_answers(golden):
    """friture/test/test_pitch_tracker.py:42-52 expects 3000 Hz from a 32-point frame; the reference's
    current estimator gates that frame out (nan) and its ungated estimate is 74.5 Hz — recorded as the
    reference behaves today, and reproduced by the oracle."""
    g = golden("pitch")
    freqs, kernels = dsp.swipe_tables()
    frame = g["kat32_frame"]
    f0, conf, db = dsp.pitch_candidate(frame, dsp.hann_symmetric(32), freqs, kernels)
    assert f0 == g["kat32_raw"][0]
    assert np.isnan(g["kat32_reference_today"][0]) and np.isnan(dsp.PitchGate().step(f0, conf, db))
