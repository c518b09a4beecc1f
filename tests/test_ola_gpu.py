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


@pytest.mark.parametrize("bpo", [1, 3, 24])
def test_batched_bank_equals_blockwise_reference(hip, bpo):
    """FirBank on a long input = the reference fed in 1024-sample blocks (oracle OlaBank, pinned to
    Octave_Filters.filter by make_golden): band signals to 1e-11 of each band's maximum, over two calls (the 511-sample
    tails cross from a batched call into a batched call and into a streaming call), several channels."""
    from friture_amd.filter import FirBank
    C, n1, n2 = 2, 9 * 1024, 4 * 1024
    x = np.stack([synth("noise", n1 + n2 + 700, 31 + c).astype(np.float64) for c in range(C)])
    bank = FirBank(bpo, C)
    refs = [dsp.OlaBank(bpo) for _ in range(C)]

    def ref_blocks(c, seg):
        parts = None
        for b in range(0, seg.shape[0], 1024):
            y, dec = refs[c].filter(seg[b:b + 1024])
            parts = [[v] for v in y] if parts is None else [p + [v] for p, v in zip(parts, y)]
        return [np.concatenate(p) for p in parts], dec

    for seg in (x[:, :n1], x[:, n1:n1 + n2], x[:, n1 + n2:n1 + n2 + 700]):
        got, dec = bank.filter(seg)
        for c in range(C):
            want, dref = ref_blocks(c, seg[c])
            assert dec == dref
            for k in range(9 * bpo):
                assert got[c][k].shape == want[k].shape, (k, got[c][k].shape, want[k].shape)
                scale = max(np.max(np.abs(want[k])), 1e-3)
                assert np.max(np.abs(got[c][k] - want[k])) <= 1e-11 * scale, (c, k)


def test_batched_bank_ragged_and_short_batches(hip):
    """Batch lengths that are not multiples of 1024 or of the kernel's 3072-sample tile, odd lengths, and a batch
    shorter than the tails it inherits (1025 .. 8191 samples), chained on one bank."""
    from friture_amd.filter import FirBank
    bank, ref = FirBank(3, 1), dsp.OlaBank(3)
    x = synth("chirp", 40000, 5).astype(np.float64)
    pos = 0
    for n in (3073, 1025, 7777, 1024, 300, 6145, 2049):
        seg = x[pos:pos + n]
        pos += n
        got, _ = bank.filter(seg[None, :])
        parts = None
        for b in range(0, n, 1024):
            y, _ = ref.filter(seg[b:b + 1024])
            parts = [[v] for v in y] if parts is None else [p + [v] for p, v in zip(parts, y)]
        for k in range(27):
            want = np.concatenate(parts[k])
            assert got[0][k].shape == want.shape
            assert np.max(np.abs(got[0][k] - want)) <= 1e-11 * max(np.max(np.abs(want)), 1e-3), (n, k)


@pytest.mark.parametrize("bpo,block", [(3, 1024), (24, 1024), (3, 256), (3, 512)])
def test_batched_bank_energies(hip, bpo, block):
    """frt_octbank_energies on the FIR bank: smoothed band energies per block (octavespectrum.py:101-121) against the
    oracle's OlaBank + exp smoothing fed block by block; 1e-5 is the north star's band-energy tolerance (measured 1e-12),
    state carried across two calls."""
    from friture_amd.filter import FirBank
    C, nb = 2, 12
    x = np.stack([synth("noise", 2 * nb * block, 77 + c) for c in range(C)])
    alphas, kernels = dsp.band_smoothing_setup(bpo, 0.125)
    bank = FirBank(bpo, C)
    got = np.concatenate([bank.energies(x[:, :nb * block], block, alphas), bank.energies(x[:, nb * block:], block, alphas)], axis=1)
    assert got.shape == (C, 2 * nb, 9 * bpo)
    for c in range(C):
        ref, prev = dsp.OlaBank(bpo), [0.0] * (9 * bpo)
        for b in range(2 * nb):
            y, _ = ref.filter(x[c, b * block:(b + 1) * block].astype(np.float64))
            prev = dsp.band_energies(y, kernels, alphas, prev)
            want = np.array(prev)
            # (a band that has not responded yet holds rounding noise of the transform, 1e-30 of the loudest band)
            assert np.all(np.abs(got[c, b] - want) <= 1e-5 * want + 1e-20 * want.max()), (c, b)


@pytest.mark.parametrize("bpo", [1, 3, 24])
def test_chunk_energies_against_oracle(hip, bpo):
    """One chunk = one energy block (the octave-spectrum widget's handler): the two-launch running-convolution path of
    frt_octbank_energies against the oracle's OlaBank + exp smoothing chunk by chunk (1e-5, the north star's band-energy
    tolerance; float32 out), for the widget's 512, the extremes 1 and 1024, odd and ragged lengths, two channels; and
    interleaved with a batched call (ola_pair_kernel): both paths carry the same tails."""
    from friture_amd.filter import FirBank
    C = 2
    sizes = [512, 512, 1, 1024, 333, 7, 512, 2, 640, 1023, 512, 100, 512]
    total = sum(sizes) + 4096
    x = np.stack([synth("noise", total, 300 + c) for c in range(C)])
    alphas, kernels = dsp.band_smoothing_setup(bpo, 0.125)
    bank = FirBank(bpo, C)
    refs = [(dsp.OlaBank(bpo), [0.0] * (9 * bpo)) for _ in range(C)]
    pos = 0

    def oracle(c, chunk):
        ref, prev = refs[c]
        y, _ = ref.filter(chunk.astype(np.float64))
        prev = dsp.band_energies(y, kernels, alphas, prev)
        refs[c] = (ref, prev)
        return np.array(prev)

    for i, n in enumerate(sizes):
        chunk = x[:, pos:pos + n]
        pos += n
        got = bank.energies(chunk, n, alphas)[:, 0]
        for c in range(C):
            want = oracle(c, chunk[c])
            assert np.all(np.abs(got[c] - want) <= 1e-5 * want + 1e-20 * want.max()), (bpo, i, n, c)
        if i == 6:
            # a batched call in between (four blocks of 1024): the transform kernels read and leave the same tails
            blk = x[:, pos:pos + 4096]
            pos += 4096
            gb = bank.energies(blk, 1024, alphas)
            for c in range(C):
                for b in range(4):
                    want = oracle(c, blk[c, b * 1024:(b + 1) * 1024])
                assert np.all(np.abs(gb[c, 3] - want) <= 1e-5 * want + 1e-20 * want.max()), (bpo, c)


def test_chunk_energies_device_pointers_and_db(hip):
    """The same path on device pointers (a torch CUDA chunk in, a torch CUDA band vector out) with the dB + weighting
    epilogue: equal to the host-buffer call."""
    torch = pytest.importorskip("torch")
    from friture_amd.filter import FirBank
    bpo, C = 3, 3
    x = np.stack([synth("noise", 4 * 512, 500 + c) for c in range(C)])
    alphas, _ = dsp.band_smoothing_setup(bpo, 0.125)
    w = np.linspace(-3.0, 1.0, 9 * bpo)
    a, b = FirBank(bpo, C), FirBank(bpo, C)
    for i in range(4):
        chunk = np.ascontiguousarray(x[:, i * 512:(i + 1) * 512], np.float32)
        host = a.energies(chunk, 512, alphas, weight_db=w, as_db=True)
        dev = b.energies(torch.from_numpy(chunk).cuda(), 512, alphas, weight_db=w, as_db=True)
        torch.cuda.synchronize()
        assert np.array_equal(host, dev.cpu().numpy())


@pytest.mark.parametrize("bpo", [3, 24])
def test_chunk_filter_equals_transform_path(hip, bpo, option):
    """Octave_Filters.filter on one block: the running-convolution launches (default for blocks of up to 1024 samples) against
    the per-stage transform launches (frt_set_option("ola_chunk_kernels", 0)) — the same sums in another order: 1e-12 of each band's scale,
    equal shapes and decimation factors, tails shared when the two alternate, two channels' worth of state kept apart."""
    from friture_amd.octavefilters import Octave_Filters
    a, b, mixed = Octave_Filters(bpo), Octave_Filters(bpo), Octave_Filters(bpo)
    x = synth("noise", 9000, 21).astype(np.float64)
    pos = 0
    for i, n in enumerate((512, 512, 1, 1024, 333, 7, 512, 1023, 640, 2, 512)):
        chunk = x[pos:pos + n]
        pos += n
        ya, da = a.filter(chunk)
        option("ola_chunk_kernels", 0)
        yb, db = b.filter(chunk)
        if i % 2:
            ym, _ = mixed.filter(chunk)
        option("ola_chunk_kernels", -1)
        if not i % 2:
            ym, _ = mixed.filter(chunk)
        assert da == db and len(ya) == len(yb) == 9 * bpo
        for k in range(9 * bpo):
            assert ya[k].shape == yb[k].shape
            scale = max(np.max(np.abs(yb[k])), 1e-3)
            assert np.max(np.abs(ya[k] - yb[k])) <= 1e-12 * scale, (bpo, n, k)
            assert np.max(np.abs(ym[k] - yb[k])) <= 1e-12 * scale, (bpo, n, k)


@pytest.mark.parametrize("bpo", [6, 12, 24])
def test_deferred_band_filters_equal_the_per_stage_launches(hip, option, bpo):
    """From 6 bands per octave the batched FFT bank runs the low-rate stages' decimators ahead and their band filters in ONE deferred
    launch (ola_pair_multi_kernel, sub-ranges of the filters per workgroup); option "ola_defer" = 0 keeps a launch per stage.  The same
    three consecutive calls (the 511-sample tails carried from call to call, band signals and energies) through both: 1e-12 of the
    signal's scale — the two forms differ only in which workgroup repeats a forward transform."""
    from friture_amd.filter import FirBank
    C, n = 2, 8 * 1024
    x = np.stack([synth("noise", 3 * n, 500 + c) for c in range(C)]).astype(np.float64)
    alphas, _ = dsp.band_smoothing_setup(bpo, 0.125)
    got = {}
    for defer in (-1, 0):
        option("ola_defer", defer)
        bank, ebank = FirBank(bpo, C), FirBank(bpo, C)
        ys = [bank.filter(x[:, i * n:(i + 1) * n])[0] for i in range(3)]        # [channel][band] arrays
        es = [ebank.energies(x[:, i * n:(i + 1) * n].astype(np.float32), 1024, alphas) for i in range(3)]
        got[defer] = (ys, es)
    for i in range(3):
        ya, yb = got[-1][0][i], got[0][0][i]
        for c in range(C):
            for a, b in zip(ya[c], yb[c]):
                assert a.shape == b.shape and np.max(np.abs(a - b)) <= 1e-12 * max(1.0, np.max(np.abs(b))), (i, c)
        ea, eb = got[-1][1][i].astype(np.float64), got[0][1][i].astype(np.float64)
        assert np.all(np.abs(ea - eb) <= 1e-6 * np.abs(eb) + 1e-20), i          # (float32 outputs of float64 energies equal to 1e-12)
