"""K3 parity: the FFT overlap-add octave bank (Octave_Filters.filter) against golden vectors
recorded from the reference and against the oracle.  float64 throughout; different FFT factorisation
than pocketfft, so agreement is to rounding: 1e-11 of each band's maximum, band energies 1e-10."""
import numpy as np
import pytest

from conftest import synth
from oracle import dsp

pytestmark = pytest.mark.gpu


def f64(x):
    return np.asarray(x, np.float32).astype(np.float64)


@pytest.mark.parametrize("bpo", [1, 3, 6, 12, 24])
def test_ola_bank_against_golden(golden, hip, bpo):
    from friture_amd.octavefilters import NOCTAVE, Octave_Filters
    g = golden("ola")
    of = Octave_Filters(bpo)
    assert NOCTAVE == 9 and of.nbands == 9 * bpo and of.FIR_LENGTH == 512
    x = f64(g[f"ola{bpo}_x"])
    pos = 0
    for blk, n in enumerate([1024, 512, 1024]):
        y, dec = of.filter(x[pos:pos + n])
        pos += n
        e = np.array([np.sum(v ** 2) for v in y])
        assert np.max(np.abs(e / g[f"ola{bpo}_energy_{blk}"] - 1)) < 1e-10
        if bpo == 3:
            for k in range(27):
                ref = g[f"ola3_y_{blk}_{k}"]
                assert y[k].shape == ref.shape and np.max(np.abs(y[k] - ref)) <= 1e-11 * np.max(np.abs(ref)), (blk, k)
    assert np.array_equal(np.array(dec), g[f"ola{bpo}_dec"]) and dec == of.get_decs()
    # band tables (O3)
    assert np.array_equal(of.fi, g[f"bands{bpo}_fi"]) and np.array_equal(of.flow, g[f"bands{bpo}_flow"])
    assert np.array_equal(of.fhigh, g[f"bands{bpo}_fhigh"])
    for got, want in zip((of.A, of.B, of.C), (g[f"bands{bpo}_A"], g[f"bands{bpo}_B"], g[f"bands{bpo}_C"])):
        assert np.array_equal(got, want)
    assert of.f_nominal == list(g[f"bands{bpo}_nominal"])


def test_ola_bank_streaming_small_and_odd_blocks(hip):
    """512-sample chunks (the widget's real feed), a 100-sample and an odd-length block: pending tails
    longer than the block are carried by addition (filter.py:235-245)."""
    from friture_amd.octavefilters import Octave_Filters
    of = Octave_Filters(3)
    ref = dsp.OlaBank(3)
    x = synth("noise", 6000, 4).astype(np.float64)
    pos = 0
    for n in (512, 512, 100, 1024, 333, 2, 512):
        y, dec = of.filter(x[pos:pos + n])
        yr, dr = ref.filter(x[pos:pos + n])
        pos += n
        assert dec == dr
        for k in range(27):
            assert y[k].shape == yr[k].shape
            scale = max(np.max(np.abs(yr[k])), 1e-3)
            assert np.max(np.abs(y[k] - yr[k])) <= 1e-11 * scale, (n, k)
    of.reset()
    ref.reset()
    y, _ = of.filter(x[:1024])
    yr, _ = ref.filter(x[:1024])
    assert all(np.max(np.abs(a - b)) <= 1e-11 * max(np.max(np.abs(b)), 1e-3) for a, b in zip(y, yr))
    with pytest.raises(Exception, match="too small"):
        of.filter(np.zeros(0))
    with pytest.raises(Exception, match="Unknown bandsperoctave"):
        of.setbandsperoctave(5)


def test_upstream_energy_property_on_gpu(hip):
    """friture/test/test_octave_filters.py:37-61 replayed with both banks on the GPU: band energies of
    the FFT bank within 5 % of the exact IIR bank over 8 blocks of default_rng(42) noise."""
    from friture_amd.filter import octave_filter_bank_decimation, octave_filter_bank_decimation_filtic
    from friture_amd.octavefilters import Octave_Filters
    for bpo in (1, 6):
        of = Octave_Filters(bpo)
        zs = octave_filter_bank_decimation_filtic(of.bdec, of.adec, of.boct, of.aoct)
        x = np.random.default_rng(42).standard_normal(8 * 1024)
        e_f, e_i = np.zeros(9 * bpo), np.zeros(9 * bpo)
        for b in range(8):
            blk = x[b * 1024:(b + 1) * 1024]
            yf, _ = of.filter(blk)
            yi, _, zs = octave_filter_bank_decimation(of.bdec, of.adec, of.boct, of.aoct, blk, zs)
            if b >= 2:
                e_f += [np.sum(v ** 2) for v in yf]
                e_i += [np.sum(v ** 2) for v in yi]
        assert np.all(np.abs(e_f / e_i - 1) < 0.05)
