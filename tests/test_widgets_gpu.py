"""Widget-level chains (the P3 drivers and the §8f "next" rows) against the oracle: the Qt-free engines
reproduce what the widgets' handle_new_data slots compute, chunk by chunk."""
import numpy as np
import pytest

from conftest import rel_max, synth
from oracle import dsp

pytestmark = pytest.mark.gpu


def test_spectrum_post_against_golden(golden, hip):
    import ctypes

    from friture_amd import _lib
    g = golden("spectrum")
    spn = g["spn"]                                   # (bins, frames)
    psd = np.ascontiguousarray(spn.T)
    alpha = float(g["kern_alpha"])
    kern = dsp.smoothing_kernel(alpha, 8192)
    prev = np.zeros(513)
    sm, db = np.empty(513), np.empty(513)
    peak, pitch = ctypes.c_int(), ctypes.c_int()
    w = np.ascontiguousarray(g["weight"])
    _lib.check(hip.frt_spectrum_post(psd.ctypes.data, 0, psd.shape[0], 513, 513, kern.ctypes.data, 8192, alpha,
                                     prev.ctypes.data, w.ctypes.data, None, sm.ctypes.data, db.ctypes.data,
                                     ctypes.byref(peak), ctypes.byref(pitch)))
    assert rel_max(sm, g["smoothed"]) < 1e-13 and np.max(np.abs(db - g["db"])) < 1e-9
    assert peak.value == int(g["peak_index"]) and pitch.value == int(g["pitch_index"])
    # float32 PSD slab (what the batch STFT engine leaves in HBM) and the dual-channel ratio
    psd32 = psd.astype(np.float32)
    ref = dsp.spectrum_readout(psd32.astype(np.float64).T, kern, alpha, prev, None, np.arange(513.0), ref_smoothed=g["smoothed"])
    _lib.check(hip.frt_spectrum_post(psd32.ctypes.data, 1, psd.shape[0], 513, 513, kern.ctypes.data, 8192, alpha,
                                     prev.ctypes.data, None, np.ascontiguousarray(g["smoothed"]).ctypes.data, sm.ctypes.data,
                                     db.ctypes.data, ctypes.byref(peak), ctypes.byref(pitch)))
    assert rel_max(sm, ref["smoothed"]) < 1e-13 and np.max(np.abs(db - ref["db"])) < 1e-9
    assert peak.value == ref["peak_index"] and pitch.value == ref["pitch_index"]


def test_spectrum_analyzer_stream(hip):
    """512-sample chunks through SpectrumAnalyzer vs the widget body restated with the oracle."""
    from friture_amd.spectrum import SpectrumAnalyzer
    n_fft, overlap = 1024, 0.75
    hop = 256
    sa = SpectrumAnalyzer(n_fft, overlap, weighting=1, response_time=0.025)
    x = (0.5 * np.sin(2 * np.pi * 440 * np.arange(512 * 20) / 48000) + 0.01 * synth("noise", 512 * 20, 2)).astype(np.float32)
    ring, old_index = dsp.MirrorRing(), 0
    prev = np.zeros(513)
    w = dsp.weighting_curves(dsp.frequency_axis(n_fft))[0]
    last = None
    for c in range(20):
        chunk = x[None, c * 512:(c + 1) * 512].astype(np.float64)
        got = sa.handle_new_data(chunk)
        ring.push(chunk)
        realizable = int(np.floor((ring.offset - old_index) / (n_fft * (1 - overlap))))
        if realizable > 0:
            cols = []
            for _ in range(realizable):
                cols.append(dsp.psd_frame(ring.data_indexed(old_index, n_fft)[0], dsp.hann_symmetric(n_fft)))
                old_index += hop
            last = dsp.spectrum_readout(np.stack(cols, axis=1), sa.kernel, sa.alpha, prev, w, dsp.frequency_axis(n_fft))
            prev = last["smoothed"]
            assert got is not None
            assert np.max(np.abs(got[1] - last["db"])) < 1e-8
            assert got[2] == last["fmax"] and got[3] == last["fpitch"]
        else:
            assert got is None
    assert abs(last["fmax"] - 440) < 48000 / n_fft


def test_stft_stream_equals_batch(hip):
    import torch

    from friture_amd.stft import StftEngine
    from friture_amd.stream import StftStream
    x = np.stack([synth("noise", 40000, 1), synth("chirp", 40000, 2)])
    st = StftStream(1024, 256, 2, max_chunk=512)
    parts = []
    pos = 0
    for n in [512] * 30 + [100, 7, 3000, 512, 512]:
        parts.append(st.push(x[:, pos:pos + n]))
        pos += n
    torch.cuda.synchronize()
    got = torch.cat(parts, dim=1).cpu().numpy()
    want = StftEngine(1024, 256, 2, 32).psd(x[:, :pos])
    assert got.shape == want.shape and np.array_equal(got, want)


def test_octave_spectrum_chain(hip):
    from friture_amd.octavespectrum import OctaveSpectrum
    osp = OctaveSpectrum(3, weighting=1, response_time=0.125)
    ref_bank = dsp.OlaBank(3)
    alphas, kernels = dsp.band_smoothing_setup(3, 0.125)
    prev = [0.0] * 27
    x = synth("noise", 512 * 12, 9).astype(np.float64)
    fi, _, _ = dsp.octave_frequencies(27, 3)
    A = dsp.band_weighting(fi)[0]
    for c in range(12):
        chunk = x[None, c * 512:(c + 1) * 512]
        got = osp.handle_new_data(chunk)
        y, _ = ref_bank.filter(chunk[0])
        prev = dsp.band_energies(y, kernels, alphas, prev)
        ref_db = dsp.band_db(prev, A)
        assert np.max(np.abs(got[3] - ref_db)) < 1e-7
    assert got[2] == osp.filters.f_nominal and len(got[2]) == 27


def test_spectrogram_chain(hip):
    """ring -> STFT(f64) -> dB/normalise -> freq resample -> time resample -> colour, chunked."""
    from friture_amd.spectrogram import Spectrogram
    sg = Spectrogram(fft_size=1024, weighting=1, screen_width=600, screen_height=120, timerange_s=2.0)
    x = synth("chirp", 512 * 16, 4).astype(np.float64)
    lut = dsp.colour_lut(dsp.cmrmap())
    w = dsp.weighting_curves(dsp.frequency_axis(1024))[0]
    tg = dsp.frequency_targets("mel", 20.0, 20000.0, 120)
    ratio = sg.sfft_rate_frac / (__import__("fractions").Fraction(600, 2000))
    tr = dsp.TimeResampler(sg.sfft_rate_frac, __import__("fractions").Fraction(600, 2000), 120)
    ring, old_index = dsp.MirrorRing(), 0
    total_px, mismatched, unaccounted = 0, 0, 0
    for c in range(16):
        chunk = x[None, c * 512:(c + 1) * 512]
        got = sg.handle_new_data(chunk)
        ring.push(chunk)
        realizable = int(np.floor((ring.offset - old_index) / 256.0))
        if realizable <= 0:
            assert got is None
            continue
        cols = []
        for _ in range(realizable):
            cols.append(dsp.psd_frame(ring.data_indexed(old_index, 1024)[0], dsp.hann_symmetric(1024)))
            old_index += 256
        norm = dsp.normalise(dsp.log_spectrum(np.stack(cols, axis=1)) + w[:, None], -140.0, 0.0)
        vals = tr.push(dsp.frequency_resample(tg, dsp.frequency_axis(1024), norm))
        want = dsp.colour_pixels(lut, vals)
        assert got.shape == want.shape and got.dtype == np.uint32
        total_px += want.size
        bad = got != want
        mismatched += int(np.sum(bad))
        # the float64 GPU transform differs from pocketfft by ~1e-16 of the frame maximum: through dB and normalisation that
        # is < 1e-12 index units for every bin above the 1e-30 floor, so a pixel may differ only where the value the colour
        # is taken from sits within 1e-9 of an index edge — each differing pixel is checked for that, not counted
        q = np.clip(vals, 0.0, 1.0) * 255
        unaccounted += int(np.sum(bad & (np.abs(q - np.rint(q)) > 1e-9)))
    assert total_px > 0 and unaccounted == 0 and mismatched <= 5, (mismatched, unaccounted, total_px)


def test_spectrogram_stream_equals_host_chain(hip):
    """SpectrogramStream (one device-resident object: ring, spectra, frequency map, carried column, LUT in HBM) against
    the block-by-block chain of Spectrogram (three host-staged pipeline blocks): the same float64 operations in the same
    order, so the SAME pixels — bit for bit, after the widget's flip of the frequency axis — over ragged chunks, more
    samples than the ring holds, a resize of the plot, a change of the time range and of the frequency scale."""
    from fractions import Fraction

    from friture_amd.plotting import frequency_scales as fs
    from friture_amd.spectrogram import Spectrogram, SpectrogramStream
    kw = dict(fft_size=1024, overlap=Fraction(3, 4), weighting=1, screen_width=500, screen_height=137, timerange_s=3.0)
    host, dev = Spectrogram(**kw), SpectrogramStream(ring_length=6000, **kw)
    x = np.concatenate([synth("chirp", 30000, 4), synth("noise", 30000, 5), synth("tone", 20000, 6)]).astype(np.float64)
    pos, blocks, cols = 0, 0, 0
    for step, n in enumerate([512] * 20 + [100, 7, 3000, 512, 1, 1, 2048] + [512] * 60 + [1500, 37] * 8):
        if step == 30:
            host.screen_height = dev.screen_height = 211                    # window resized: Fourier-resampled carried column
        if step == 45:
            host.timerange_s = dev.timerange_s = 1.0                        # other pixel rate: indices restart
        if step == 60:
            host.frequency_resampler.setfreqscale(fs.Logarithmic)
            dev.scale = fs.Logarithmic
        if step == 75:
            host.screen_height = dev.screen_height = 64
        chunk = x[None, pos:pos + n]
        pos += n
        a, b = host.handle_new_data(chunk), dev.handle_new_data(chunk)
        assert (a is None) == (b is None), step
        if a is not None:
            assert b.shape == a.shape and b.dtype == np.uint32, (step, a.shape, b.shape)
            assert np.array_equal(b, a[::-1, :]), step
            blocks += 1
            cols += a.shape[1]
    assert blocks > 50 and cols > 300 and pos > 6000 * 9          # the 6000-sample ring wrapped nine times


def test_spectrogram_stream_small_chunks_large_frames(hip):
    """fft_size 16384 fed in 512-sample chunks back to back: seven of eight pushes complete no frame and return without
    waiting for their upload, so the staging buffers are reused immediately (the copy of chunk k must have
    left them before chunk k + 1 is written there).  Same pixels as the block-by-block chain, and as ONE large push."""
    from fractions import Fraction

    from friture_amd.spectrogram import Spectrogram, SpectrogramStream
    kw = dict(fft_size=16384, overlap=Fraction(1, 2), weighting=1, screen_width=300, screen_height=97, timerange_s=2.0)
    host, dev, big = Spectrogram(**kw), SpectrogramStream(**kw), SpectrogramStream(**kw)
    x = synth("chirp", 512 * 200, 21).astype(np.float64)
    got_host, got_dev = [], []
    for k in range(200):
        chunk = x[None, 512 * k:512 * (k + 1)]
        a, b = host.handle_new_data(chunk), dev.handle_new_data(chunk)
        assert (a is None) == (b is None), k
        if a is not None:
            got_host.append(a[::-1, :])
            got_dev.append(b)
    assert len(got_dev) >= 10
    assert np.array_equal(np.concatenate(got_dev, axis=1), np.concatenate(got_host, axis=1))
    # one push of everything: longer than the ring (65536 samples) -> the ring grows like the reference's (ringbuffer.py:102-130)
    c = big.handle_new_data(x[None, :])
    assert c is not None and np.array_equal(c, np.concatenate(got_dev, axis=1))
    # and the grown object keeps streaming: the same continuation from both
    y = synth("noise", 16384 * 2, 22).astype(np.float64)
    d1 = [dev.handle_new_data(y[None, i:i + 4096]) for i in range(0, len(y), 4096)]
    d2 = [big.handle_new_data(y[None, i:i + 4096]) for i in range(0, len(y), 4096)]
    for u, v in zip(d1, d2):
        assert (u is None) == (v is None)
        if u is not None:
            assert np.array_equal(u, v)


def test_spectrogram_stream_error_leaves_state(hip):
    """A push whose pixel columns do not fit the caller's block fails BEFORE any state changes: the same chunk pushed again
    with room gives what an undisturbed object gives."""
    import ctypes as ct
    from fractions import Fraction

    from friture_amd import _lib
    from friture_amd.spectrogram import SpectrogramStream
    kw = dict(fft_size=1024, overlap=Fraction(3, 4), weighting=0, screen_width=800, screen_height=64, timerange_s=0.5)
    a, b = SpectrogramStream(**kw), SpectrogramStream(**kw)
    x = synth("noise", 40000, 31).astype(np.float64)
    ra = a.handle_new_data(x[None, :20000])
    a._sync_screen()
    tiny = np.zeros((64, 2), np.uint32)
    ncols, nfr = ct.c_int(0), ct.c_int(0)
    chunk = np.ascontiguousarray(x[20000:40000])
    rc = a._lib.frt_specgram_push(a._h, chunk.ctypes.data, len(chunk), tiny.ctypes.data, 2, ct.byref(ncols), ct.byref(nfr))
    assert rc != 0 and b"pixel columns" in a._lib.frt_last_error()
    rb = b.handle_new_data(x[None, :20000])
    assert np.array_equal(ra, rb)
    assert np.array_equal(a.handle_new_data(x[None, 20000:40000]), b.handle_new_data(x[None, 20000:40000]))


def test_octave_spectrum_stream_equals_host_chain(hip):
    """OctaveSpectrumStream (FIR bank tails, smoothed energies, weighting all on the device; float32 band vector out)
    against OctaveSpectrum (band signals back to the host, smoothing per decimation class) over the widget's 512-sample
    chunks and ragged ones: 1e-5 on the energies = 4.3e-5 dB."""
    from friture_amd.octavespectrum import OctaveSpectrum, OctaveSpectrumStream
    for bpo in (3, 12):
        a, b = OctaveSpectrum(bpo, weighting=1, response_time=0.125), OctaveSpectrumStream(bpo, weighting=1, response_time=0.125)
        x = synth("noise", 20000, 40 + bpo).astype(np.float64)         # float32-representable, like captured audio
        pos = 0
        for n in [512] * 12 + [100, 333, 1024, 7, 512, 1, 640]:
            chunk = x[None, pos:pos + n]
            pos += n
            ra, rb = a.handle_new_data(chunk), b.handle_new_data(chunk)
            assert ra[2] == rb[2] and np.array_equal(ra[0], rb[0])
            assert np.max(np.abs(ra[3] - rb[3])) <= 4.4e-5, (bpo, n, float(np.max(np.abs(ra[3] - rb[3]))))


@pytest.mark.parametrize("dual", [False, True])
def test_spectrum_analyzer_stream_equals_host_chain(hip, dual):
    """SpectrumAnalyzerStream (device ring, smoothed spectra resident in HBM) against SpectrumAnalyzer chunk after chunk:
    identical dB spectra, peak and pitch, single and dual channel (the dual-channel dB is the ratio to channel 1's
    smoothed spectrum, spectrum.py:160-175)."""
    from friture_amd.spectrum import SpectrumAnalyzer, SpectrumAnalyzerStream
    kw = dict(fft_size=2048, overlap=0.75, weighting=2, response_time=0.05, dual_channels=dual)
    a, b = SpectrumAnalyzer(**kw), SpectrumAnalyzerStream(**kw)
    rows = 2 if dual else 1
    x = np.stack([synth("tone", 30000, 3), synth("noise", 30000, 8)])[:rows].astype(np.float64)
    pos, got_any = 0, 0
    for n in [512] * 20 + [100, 3000, 7, 512, 2048] + [512] * 20:
        chunk = x[:, pos:pos + n]
        pos += n
        ra, rb = a.handle_new_data(chunk), b.handle_new_data(chunk)
        assert (ra is None) == (rb is None)
        if ra is not None:
            got_any += 1
            assert np.array_equal(ra[1], rb[1]) and ra[2] == rb[2] and ra[3] == rb[3]
    assert got_any > 30
