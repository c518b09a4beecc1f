"""K6 / P5-P8 parity: screen-space resamplers, colour map and block smoothing against golden vectors
recorded from the reference.  float64 with the reference's operation order: bit-exact, except the
smoothing dot product (BLAS summation order is unspecified): 1e-13."""
import numpy as np
import pytest

from oracle import dsp

pytestmark = pytest.mark.gpu


def test_frequency_resampler(golden, hip):
    from friture_amd.plotting import frequency_scales as fs
    from friture_amd.signal.frequency_resampler import Frequency_Resampler
    g = golden("pipeline")
    for name, scale in [("linear", fs.Linear), ("log", fs.Logarithmic), ("mel", fs.Mel), ("erb", fs.Erb), ("octave", fs.Octave)]:
        fr = Frequency_Resampler(scale, 20.0, 20000.0, 100)
        fr.setfreq(g["freq"])
        assert np.array_equal(fr.xscaled, g[f"fr_{name}_targets"])
        assert np.array_equal(fr.push(g["norm"]), g[f"fr_{name}"])
    # targets outside the table clamp to the edge bins, exact hits return the bin (numpy.interp)
    fr = Frequency_Resampler(fs.Linear, 0.0, 24000.0, 513)
    fr.setfreq(g["freq"])
    assert np.array_equal(fr.push(g["norm"]), g["norm"])
    fr.setfreqrange(-100.0, 30000.0)
    out = fr.push(g["norm"])
    assert np.array_equal(out[0], g["norm"][0]) and np.array_equal(out[-1], g["norm"][-1])
    assert np.array_equal(out, dsp.frequency_resample(fr.xscaled, g["freq"], g["norm"]))


def test_time_resampler(golden, hip):
    from friture_amd.signal.online_linear_2D_resampler import Online_Linear_2D_resampler
    g = golden("pipeline")
    for tag, (L, M) in {"down": (25, 16), "up": (3, 7)}.items():
        tr = Online_Linear_2D_resampler(L, M, 100)
        assert np.array_equal(tr.push(g["fr_mel"][:, :5]), g[f"tr_{tag}_a"])
        assert np.array_equal(tr.push(g["fr_mel"][:, 5:]), g[f"tr_{tag}_b"])
    # column-at-a-time pushes give the same stream as one big push
    one = Online_Linear_2D_resampler(25, 16, 100)
    ref = dsp.TimeResampler(25, 16, 100)
    parts = [one.push(g["fr_mel"][:, j:j + 1]) for j in range(12)]
    assert np.array_equal(np.concatenate(parts, axis=1), ref.push(g["fr_mel"]))


def test_time_resampler_across_a_resize(golden, hip):
    """set_height (online_linear_2D_resampler.py:45-55): a push of another height Fourier-resamples the carried column
    (scipy_resample.py:51-141).  Against the reference's own outputs over 100 -> 137 -> 64 rows; direct DFT sums on the
    device against pocketfft: 1e-12 of the column maximum."""
    from friture_amd.signal.online_linear_2D_resampler import Online_Linear_2D_resampler
    from friture_amd.signal.scipy_resample import resample
    g = golden("pipeline")
    col = g["fr_mel"][:, 7]
    for h in (137, 64, 100, 211):
        ref = g[f"fourier_100_{h}"]
        got = resample(col, h)
        assert got.shape == ref.shape and np.max(np.abs(got - ref)) <= 1e-12 * np.max(np.abs(ref)), h
    # awkward lengths: primes both ways, a single row, 2-D input along axis 0
    rng = np.random.default_rng(3)
    for n, m in [(997, 1009), (1009, 463), (1, 5), (2, 3), (3, 2), (480, 1080), (1080, 479)]:
        x = rng.standard_normal((n, 3))
        ref = dsp.fourier_resample(x, m)
        got = resample(x, m)
        assert got.shape == ref.shape and np.max(np.abs(got - ref)) <= 1e-12 * max(np.max(np.abs(ref)), 1e-300), (n, m)
    with pytest.raises(ValueError):
        resample(np.ones(7), 1)                      # the reference raises here too (its negative-frequency slice)
    tr = Online_Linear_2D_resampler(25, 16, 100)
    for i in range(3):
        out = tr.push(g[f"tr_resize_in_{i}"])
        ref = g[f"tr_resize_out_{i}"]
        assert out.shape == ref.shape
        assert np.max(np.abs(out - ref)) <= 1e-12 * np.max(np.abs(ref)), i


def test_colour_transform_and_full_pipeline(golden, hip):
    from friture_amd.plotting import frequency_scales as fs
    from friture_amd.signal.color_tranform import Color_Transform
    from friture_amd.signal.frequency_resampler import Frequency_Resampler
    from friture_amd.signal.online_linear_2D_resampler import Online_Linear_2D_resampler
    from friture_amd.signal.transform_pipeline import Transform_Pipeline
    img = golden("image")
    ct = Color_Transform()
    assert np.array_equal(ct.colors, img["lut"])
    assert np.array_equal(ct.push(img["norm"]), img["image"])
    edge = np.array([[-1.0, 0.0, 1.0 / 255, 0.999999, 1.0, 7.0]])
    assert np.array_equal(ct.push(edge), dsp.colour_pixels(img["lut"], edge))
    # the three blocks chained as Spectrogram_Widget builds them (spectrogram.py:62-68)
    g = golden("pipeline")
    fr = Frequency_Resampler(fs.Mel, 20.0, 20000.0, 100)
    fr.setfreq(g["freq"])
    pipe = Transform_Pipeline([fr, Online_Linear_2D_resampler(25, 16, 100), ct])
    got = pipe.push(g["norm"])
    tr = dsp.TimeResampler(25, 16, 100)
    want = dsp.colour_pixels(img["lut"], tr.push(g["fr_mel"]))
    assert got.dtype == np.uint32 and np.array_equal(got, want)


def test_exp_smoothing(golden, hip):
    from friture_amd.signal.exp_smoothing import exp_smoothed_value, exp_smoothed_value_2d
    g = golden("exp_smoothing")
    assert abs(exp_smoothed_value(g["kern"], 0.02, g["d1"], 0.3) / float(g["r1"]) - 1) < 1e-13
    assert np.max(np.abs(exp_smoothed_value_2d(g["kern"], 0.02, g["d2"], g["prev"]) / g["r2"] - 1)) < 1e-13
    assert np.max(np.abs(exp_smoothed_value_2d(g["kern"], 0.02, g["d2"][:, :17], g["prev"]) / g["r3"] - 1)) < 1e-13
    assert exp_smoothed_value(g["kern"], 0.02, np.zeros(0), 0.7) == 0.7
    assert np.array_equal(exp_smoothed_value_2d(g["kern"], 0.02, np.zeros((5, 0)), g["prev"]), g["prev"])
    # upstream's properties (friture/test/test_exp_smoothing.py:9-44)
    row = np.random.default_rng(0).random(20)
    out = exp_smoothed_value_2d(dsp.smoothing_kernel(0.1, 32), 0.1, np.stack([row, row, 2 * row]), np.zeros(3))
    assert out[0] == out[1] and out[2] > out[0]


def test_fused_transform_pipeline_equals_block_by_block(hip):
    """Transform_Pipeline.push over the spectrogram's three blocks runs as ONE device call (frt_screen_columns); the same
    blocks pushed one after the other (the reference's reduce(), three calls) give the same pixels and leave the same state
    — over many pushes, a resize of the screen (Fourier-resampled carried column) and a change of the pixel rate."""
    from functools import reduce

    from friture_amd.plotting import frequency_scales as fs
    from friture_amd.signal.color_tranform import Color_Transform
    from friture_amd.signal.frequency_resampler import Frequency_Resampler
    from friture_amd.signal.online_linear_2D_resampler import Online_Linear_2D_resampler
    from friture_amd.signal.transform_pipeline import Transform_Pipeline

    def chain():
        fr = Frequency_Resampler(fs.Mel, 20., 20000., 137)
        fr.setfreq(np.linspace(0, 24000, 513))
        tr = Online_Linear_2D_resampler()
        tr.set_ratio(0.1875, 0.08)
        return [fr, tr, Color_Transform()]

    fused, plain = chain(), chain()
    pipe = Transform_Pipeline(fused)
    assert pipe._fusable()
    rng = np.random.default_rng(7)
    for step in range(40):
        if step == 15:
            for b in (fused, plain):
                b[0].setnsamples(211)
        if step == 25:
            for b in (fused, plain):
                b[1].set_ratio(0.1875, 0.3)
        data = rng.uniform(-0.2, 1.2, (513, int(rng.integers(1, 6))))
        got = pipe.push(data)
        want = reduce(lambda cols, stage: stage.push(cols), plain, data)
        assert got.dtype == np.uint32 and got.shape == want.shape, (step, got.shape, want.shape)
        assert np.array_equal(got, want), step
        assert np.array_equal(fused[1].old_data, plain[1].old_data)
        assert (fused[1].orig_index, fused[1].resampled_index) == (plain[1].orig_index, plain[1].resampled_index)
        if step == 30:
            # in-place changes of the blocks' arrays (no setter involved) must reach the fused call like they reach the
            # per-block pushes: a float32 LUT copy forces the fused path to keep a converted copy, then mutate it
            for b in (fused, plain):
                b[2].colors = b[2].colors.astype(np.uint64)          # not uint32: the fused path converts
        if step == 33:
            for b in (fused, plain):
                b[2].colors[:] = b[2].colors[::-1].copy()            # in place
                b[0].xscaled[:] = b[0].xscaled * 0.999               # in place


def test_exp_smoothing_groups_equal_their_own_calls(hip):
    """frt_exp_smooth_groups (the octave-spectrum widget's chunk in one launch): every row bit for bit what its own
    exp_smoothed_value_2d call returns — ragged lengths, a group longer than its kernel (previous forgotten), an empty group, the
    packed-rows form, and the squared form against squaring on the host."""
    from friture_amd.signal.exp_smoothing import exp_smoothed_value_2d, exp_smoothed_value_groups
    rng = np.random.default_rng(5)
    shapes = [(3, 512), (3, 256), (2, 64), (3, 2), (1, 1), (2, 0), (3, 40)]
    nks = [8192, 4096, 512, 64, 16, 16, 24]                    # the last group has more data than taps
    alphas = [0.01, 0.02, 0.05, 0.1, 0.2, 0.3, 0.15]
    kernels = [(1.0 - a) ** np.arange(nk - 1, -1, -1) for a, nk in zip(alphas, nks)]
    blocks = [rng.standard_normal(sh) for sh in shapes]
    prev = rng.random(sum(sh[0] for sh in shapes))
    for square in (False, True):
        got = exp_smoothed_value_groups(kernels, alphas, blocks, prev, square=square)
        pos = 0
        for k, a, b in zip(kernels, alphas, blocks):
            want = exp_smoothed_value_2d(k, a, b * b if square else b, prev[pos:pos + b.shape[0]])
            assert np.array_equal(got[pos:pos + b.shape[0]], want), (square, b.shape)
            pos += b.shape[0]
    # rows back to back in one buffer, handed over as (first row, count)
    packed = np.concatenate([b.ravel() for b in blocks])
    tuples, off = [], 0
    for (r, n), b in zip(shapes, blocks):
        tuples.append((packed[off:off + n], r) if n else b)
        off += r * n
    assert np.array_equal(exp_smoothed_value_groups(kernels, alphas, tuples, prev, square=True),
                          exp_smoothed_value_groups(kernels, alphas, blocks, prev, square=True))
