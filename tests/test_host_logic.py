"""Host-side mirrors that need no GPU: ring buffer, band labels, tables."""
import numpy as np
import pytest

from oracle import dsp


def test_ringbuffer_matches_reference(golden):
    from friture_amd.ringbuffer import RingBuffer
    g = golden("ring")
    ring = RingBuffer()
    for step in range(6):
        ring.push(g[f"blk{step}"], 0.0)
        ln = min(ring.offset, 4096)
        assert np.array_equal(ring.data_indexed(ring.offset - 100, ln - 100), g[f"win{step}"])
    assert np.array_equal(ring.data(256), ring.data_indexed(ring.offset, 256))
    assert np.array_equal(ring.data_older(100, 50), ring.data_indexed(ring.offset - 50, 100))


def test_ringbuffer_errors_and_growth():
    from friture_amd.ringbuffer import RingBuffer
    ring, ref = RingBuffer(), dsp.MirrorRing()
    rng = np.random.default_rng(0)
    for n in (300, 9900, 25000, 7):
        blk = rng.standard_normal((2, n))
        ring.push(blk, 1.0)
        ref.push(blk)
        assert ring.buffer_length == ref.buffer_length and ring.offset == ref.offset
        assert np.array_equal(ring.data(min(ring.offset, 5000)), ref.data(min(ref.offset, 5000)))
    assert ring.data_time(ring.offset - 480) == 1.0 - 0.01


def test_nominal_labels(golden):
    from friture_amd.octavefilters import nominal_labels
    g = golden("ola")
    for bpo in (1, 3, 6, 12, 24):
        fi, _, _ = dsp.octave_frequencies(9 * bpo, bpo)
        assert nominal_labels(fi, bpo) == list(g[f"bands{bpo}_nominal"])
    with pytest.raises(Exception, match="Unknown bandsperoctave"):
        nominal_labels(dsp.octave_frequencies(45, 5)[0], 5)


def test_tables_match_reference(golden):
    from friture_amd import palette, tables
    from friture_amd.filter import octave_frequencies
    g = golden("psd")
    assert np.array_equal(tables.rfft_frequencies(1024), g["N1024_freq"])
    for got, want in zip(tables.weighting_db(g["N1024_freq"], 1e-50), (g["N1024_A"], g["N1024_B"], g["N1024_C"])):
        assert np.array_equal(got, want)
    assert np.array_equal(palette.cmr_lut(), golden("image")["lut"])
    o = golden("ola")
    for bpo in (1, 3, 24):
        fi, flo, fhi = octave_frequencies(9 * bpo, bpo)
        assert np.array_equal(fi, o[f"bands{bpo}_fi"]) and np.array_equal(flo, o[f"bands{bpo}_flow"])
        assert np.array_equal(fhi, o[f"bands{bpo}_fhigh"])
        assert np.array_equal(tables.weighting_db(fi)[0], o[f"bands{bpo}_A"])


def test_time_resampler_bookkeeping_matches_oracle():
    """Online_Linear_2D_resampler.advance (plain Python floats) against the oracle's numpy restatement of the reference's index
    recurrence (online_linear_2D_resampler.py:61-97): the same pixel columns, the same float64 weights bit for bit, the same
    carried indices — for up- and down-sampling ratios and ragged pushes (no device call involved)."""
    from friture_amd.signal.online_linear_2D_resampler import Online_Linear_2D_resampler
    rng = np.random.default_rng(3)
    for L, M in ((25, 16), (1, 3), (7, 5), (375, 32), (2, 2), (1000, 3)):
        mine = object.__new__(Online_Linear_2D_resampler)               # the constructor binds the device library
        mine.resampling_ratio, mine.orig_index, mine.resampled_index = float(L) / M, 0., 0.
        ref = dsp.TimeResampler(L, M, 4)
        for _ in range(40):
            n_cols = int(rng.integers(0, 6))
            total, src, a = mine.advance(n_cols)
            # the oracle's loop, bookkeeping only
            want_total = ref._processable(n_cols)
            want_src, want_a = [], []
            for j in range(n_cols):
                ref.orig_index += 1.0
                n = ref._processable(0)
                if n > 0:
                    idx = ref.resampled_index + ref.ratio * np.arange(1, n + 1, dtype=np.float64)
                    want_a += list(ref.orig_index - idx)
                    want_src += [j] * n
                    ref.resampled_index = float(idx[-1])
            assert total == want_total and list(src) == want_src
            assert np.array_equal(a, np.array(want_a, np.float64))
            assert mine.orig_index == ref.orig_index and mine.resampled_index == ref.resampled_index


def test_frequency_scale_names_cover_the_reference_list():
    """friture/plotting/frequency_scales.py:65-293 defines Linear, Logarithmic, Mel, ERB, Octave and OctaveC; OctaveC is the
    Octave transform pair under its own name (:225-237)."""
    from friture_amd.plotting import frequency_scales as fs
    from oracle import dsp
    assert [s.NAME for s in fs.ALL] == ["Linear", "Logarithmic", "Mel", "ERB", "Octave", "OctaveC"]
    f = np.array([0.0, 20.0, 440.0, 20000.0])
    assert np.array_equal(fs.OctaveC.transform(f), fs.Octave.transform(f)) and np.array_equal(fs.OctaveC.inverse(f[1:] / 1e3), fs.Octave.inverse(f[1:] / 1e3))
    want = dsp.frequency_targets("octave", 20.0, 20000.0, 37)
    assert np.array_equal(fs.OctaveC.inverse(np.linspace(fs.OctaveC.transform(20.0), fs.OctaveC.transform(20000.0), 37)), want)


def test_backend_swap_uninstall_restores_parent_attributes():
    """install() binds the swapped modules as attributes of their parent packages (what `import pkg.sub as m` resolves);
    uninstall() must put back what was there — or remove what was not."""
    import sys
    import types

    from friture_amd import backend_swap
    pkg, sig = types.ModuleType("frt_fakepkg"), types.ModuleType("frt_fakepkg.signal")
    pkg.signal = sig
    sig.correlation = "original correlation"
    sys.modules["frt_fakepkg"], sys.modules["frt_fakepkg.signal"] = pkg, sig
    sys.modules["frt_fakepkg.signal.correlation"] = "original module"
    try:
        for _ in range(2):                                             # a second cycle must not lose the originals either
            backend_swap.install("frt_fakepkg")
            import friture_amd.audioproc
            import friture_amd.signal.correlation
            assert pkg.audioproc is friture_amd.audioproc and sig.correlation is friture_amd.signal.correlation
            assert sys.modules["frt_fakepkg.signal.correlation"] is friture_amd.signal.correlation
            backend_swap.uninstall()
            assert sig.correlation == "original correlation" and not hasattr(pkg, "audioproc")
            assert sys.modules["frt_fakepkg.signal.correlation"] == "original module" and "frt_fakepkg.audioproc" not in sys.modules
    finally:
        for k in [k for k in sys.modules if k.startswith("frt_fakepkg")]:
            sys.modules.pop(k, None)
