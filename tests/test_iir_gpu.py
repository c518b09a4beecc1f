"""K2/K4/G2 parity: exact IIR octave bank, decimation chain, band energies.

Sequential mode must be *bit-identical* to the reference (golden fixtures recorded from
friture.filter.octave_filter_bank_decimation and friture.signal.decimate) — float64, same IEEE
operations in the same order.  The time-parallel mode re-associates the linear recurrence; the 12th-order
direct-form decimator amplifies rounding by ~1e4 (its DF2T states are much larger than its output), so
the two modes agree to ~1e-10 of the *input* scale after nine cascaded stages; bands that a tone
barely excites have a small maximum, so the per-band gate is 1e-7 of the band maximum.  Band energies (the metric's band-energy vector): |E/E_ref - 1| <= 1e-5.
"""
import numpy as np
import pytest

from conftest import synth
from oracle import dsp

pytestmark = pytest.mark.gpu


def f64(x):
    return np.asarray(x, np.float32).astype(np.float64)


@pytest.fixture(scope="module")
def tabs(hip):
    return dsp.load_filter_tables()


def test_lfilter_bit_exact(golden, tabs):
    from friture_amd.signal.lfilter import lfilter_float64_1D
    x = f64(golden("iir")["x_dec"])[:700]
    y, zf = lfilter_float64_1D(tabs["bdec"], tabs["adec"], x, np.zeros(12))
    yo, zo = dsp.lfilter_df2t(tabs["bdec"], tabs["adec"], x, np.zeros(12))
    assert np.array_equal(y, yo) and np.array_equal(zf, zo)
    # restart from the carried state, band-pass filter, ragged length
    b, a = tabs["boct_3"][1], tabs["aoct_3"][1]
    y1, z1 = lfilter_float64_1D(b, a, x[:333], np.zeros(4))
    y2, z2 = lfilter_float64_1D(b, a, x[333:], z1)
    yo, zo = dsp.lfilter_df2t(b, a, x, np.zeros(4))
    assert np.array_equal(np.concatenate([y1, y2]), yo) and np.array_equal(z2, zo)
    # pure gain and empty input (lfilter.py:140-142)
    y, zf = lfilter_float64_1D(np.array([0.5]), np.array([1.0]), x[:5], np.zeros(0))
    assert np.array_equal(y, 0.5 * x[:5])
    with pytest.raises(AssertionError):
        lfilter_float64_1D(b, a[:4], x, np.zeros(4))


def test_decimate_multiple_against_golden(golden, tabs):
    from friture_amd.signal import decimate as D
    g = golden("iir")
    x = f64(g["x_dec"])
    zs = D.decimate_multiple_filtic(2, tabs["bdec"], tabs["adec"])
    for c in range(4):
        y, zs = D.decimate_multiple(2, tabs["bdec"], tabs["adec"], x[c * 512:(c + 1) * 512], zs)
        assert np.array_equal(y, g[f"dec2_{c}"])
    y, zf = D.decimate(tabs["bdec"], tabs["adec"], x[:101])
    yo, zo = dsp.decimate(tabs["bdec"], tabs["adec"], x[:101])
    assert np.array_equal(y, yo) and np.array_equal(zf, zo) and len(y) == 51
    with pytest.raises(Exception, match="too small"):
        D.decimate(tabs["bdec"], tabs["adec"], np.zeros(0))
    e, z = D.decimate_multiple(2, tabs["bdec"], tabs["adec"], np.zeros(0), None)
    assert e.size == 0 and z is None


def test_decimate_multiple_in_place_call_equals_the_staged_one(tabs):
    """decimate_multiple with explicit states: the one-call in-place path (frt_decimate_multiple_state, up to 30000 samples)
    and the staged path (set_state / frt_decimate_multiple / get_state, longer inputs) are the same sequential recurrence:
    a long signal through the staged path equals its ragged pieces through the in-place one, samples and final states bit
    for bit, for 1..4 chained stages, with and without states, and against the oracle."""
    from friture_amd.signal import decimate as D
    rng = np.random.default_rng(11)
    x = 0.3 * rng.standard_normal(70001)
    for ndec in (1, 2, 4):
        whole, zw = D.decimate_multiple(ndec, tabs["bdec"], tabs["adec"], x, D.decimate_multiple_filtic(ndec, tabs["bdec"], tabs["adec"]))
        ref, zr = dsp.decimate_multiple(ndec, tabs["bdec"], tabs["adec"], x, dsp.decimate_multiple_filtic(ndec, tabs["bdec"], tabs["adec"]))
        assert np.array_equal(whole, ref) and all(np.array_equal(a, b) for a, b in zip(zw, zr))
        # pieces whose lengths keep every stage's take-every-other phase aligned (multiples of 2^ndec), then a ragged tail
        zs = D.decimate_multiple_filtic(ndec, tabs["bdec"], tabs["adec"])
        parts, pos = [], 0
        for n in (512, 16 * 1024, 30000 - 30000 % 16, 48, 16):
            y, zs = D.decimate_multiple(ndec, tabs["bdec"], tabs["adec"], x[pos:pos + n], zs)
            parts.append(y)
            pos += n
        y, zs = D.decimate_multiple(ndec, tabs["bdec"], tabs["adec"], x[pos:], zs)      # 22000-odd samples, odd length
        parts.append(y)
        assert np.array_equal(np.concatenate(parts), whole)
        assert all(np.array_equal(a, b) for a, b in zip(zs, zw))
    y0, z0 = D.decimate_multiple(2, tabs["bdec"], tabs["adec"], x[:777], None)           # zero state, none returned
    r0, _ = dsp.decimate_multiple(2, tabs["bdec"], tabs["adec"], x[:777], dsp.decimate_multiple_filtic(2, tabs["bdec"], tabs["adec"]))
    assert z0 is None and np.array_equal(y0, r0)


def test_decimate_multiple_of_several_channels_in_one_call(tabs):
    """decimate_multiple_channels (frt_decimate_multiple_state on a handle with C channels: every channel a slot of the same
    launches) against one decimate_multiple call per channel: samples and states bit for bit, over a run of ragged chunks."""
    from friture_amd.signal import decimate as D
    rng = np.random.default_rng(17)
    for C, ndec in ((2, 2), (3, 1), (2, 4)):
        x = 0.3 * rng.standard_normal((C, 4096 + 512 + 37))
        za = [D.decimate_multiple_filtic(ndec, tabs["bdec"], tabs["adec"]) for _ in range(C)]
        zb = [D.decimate_multiple_filtic(ndec, tabs["bdec"], tabs["adec"]) for _ in range(C)]
        pos = 0
        for n in (512, 512, 2048, 1024, 512, 37):
            ya, za = D.decimate_multiple_channels(ndec, tabs["bdec"], tabs["adec"], x[:, pos:pos + n], za)
            for c in range(C):
                yb, zb[c] = D.decimate_multiple(ndec, tabs["bdec"], tabs["adec"], x[c, pos:pos + n], zb[c])
                assert np.array_equal(ya[c], yb), (C, ndec, n, c)
                assert all(np.array_equal(p, q) for p, q in zip(za[c], zb[c])), (C, ndec, n, c)
            pos += n


@pytest.mark.parametrize("bpo", [1, 3, 6, 12, 24])
def test_iir_bank_against_golden(golden, tabs, bpo):
    from friture_amd import filter as F
    g = golden("iir")
    boct, aoct = list(tabs[f"boct_{bpo}"]), list(tabs[f"aoct_{bpo}"])
    zs = F.octave_filter_bank_decimation_filtic(tabs["bdec"], tabs["adec"], boct, aoct)
    x = f64(g[f"bank{bpo}_x"])
    for blk in range(2):
        y, dec, zs = F.octave_filter_bank_decimation(tabs["bdec"], tabs["adec"], boct, aoct, x[blk * 1024:(blk + 1) * 1024], zs)
        assert np.array_equal(np.array([np.sum(v ** 2) for v in y]), g[f"bank{bpo}_energy_{blk}"])
        if bpo == 3:
            for k in range(27):
                assert np.array_equal(y[k], g[f"bank3_y_{blk}_{k}"])
    assert np.array_equal(np.array(dec), g[f"bank{bpo}_dec"])
    assert np.array_equal(np.concatenate(zs), g[f"bank{bpo}_zf"])


def test_iir_bank_ragged_blocks_and_channels(tabs):
    """odd block lengths (x[::2] keeps ceil(n/2) samples), several channels, against the oracle"""
    from friture_amd.filter import IirBank
    bpo = 3
    boct, aoct = list(tabs[f"boct_{bpo}"]), list(tabs[f"aoct_{bpo}"])
    bank = IirBank(tabs["bdec"], tabs["adec"], boct, aoct, n_channels=3)
    x = np.stack([synth("noise", 3000, 1), synth("tone", 3000, 2), synth("chirp", 3000, 3)]).astype(np.float64)
    zs = [dsp.iir_bank_filtic(tabs["bdec"], tabs["adec"], boct, aoct) for _ in range(3)]
    pos = 0
    for n in (1000, 777, 1, 1222):
        got, dec = bank.filter(x[:, pos:pos + n])
        for c in range(3):
            ref, dref, zs[c] = dsp.iir_bank(tabs["bdec"], tabs["adec"], boct, aoct, x[c, pos:pos + n], zs[c])
            assert dec == dref
            for k in range(27):
                assert np.array_equal(got[c][k], ref[k]), (n, c, k)
        pos += n
    with pytest.raises(Exception, match="too small"):
        bank.filter(np.zeros((3, 0)))


@pytest.mark.parametrize("chunk,extra", [(16384, 0), (4096, 0), (1024, 0), (3072, 0), (-2048, 0), (2048, 3), (1024, 1021)])
def test_time_parallel_mode_matches_sequential(tabs, chunk, extra):
    """`extra`: a ragged length — rows of the staged input off the 16-byte grid (the scalar-load branch of the zero-state
    kernel), a last chunk that is not whole, odd stage lengths down the decimation chain."""
    from friture_amd.filter import IirBank
    bpo = 3
    boct, aoct = list(tabs[f"boct_{bpo}"]), list(tabs[f"aoct_{bpo}"])
    n = 5 * 16384 + 4096 + extra
    x = np.stack([synth("noise", n, 5), synth("tone", n, 6)]).astype(np.float64)
    seq = IirBank(tabs["bdec"], tabs["adec"], boct, aoct, 2)
    par = IirBank(tabs["bdec"], tabs["adec"], boct, aoct, 2)
    par.set_chunk(chunk)
    # warm both with a first block so that the scan starts from a non-zero carried state
    seq.filter(x[:, :4096])
    par.filter(x[:, :4096])
    ys, _ = seq.filter(x[:, 4096:])
    yp, _ = par.filter(x[:, 4096:])
    for c in range(2):
        for k in range(27):
            # same linear recurrence in another association order: rounding errors scale with the signal that
            # drives the states (the input), not with the band's own level — a stop-band output 1e-5 below the
            # input carries them at the same absolute size
            err = np.max(np.abs(yp[c][k] - ys[c][k]))
            assert err < 1e-7 * np.max(np.abs(ys[c][k])) + 1e-10 * np.max(np.abs(x[c])), (c, k, err)
    # carried states: the 12th-order direct-form decimator amplifies rounding ~1e6-fold, so two association orders of the
    # same recurrence sit a few 1e-9 of the largest state apart (5e-9 measured with 1024-sample chunks); the outputs
    # above are held to 1e-10 of the input scale
    assert np.max(np.abs(par.get_state() - seq.get_state())) < 1e-8 * np.max(np.abs(seq.get_state()))


@pytest.mark.parametrize("bpo,chunk", [(3, 0), (3, 16384), (3, 2048), (3, 1024), (3, -2048), (24, 16384), (24, 4096)])
def test_band_energies(tabs, bpo, chunk):
    """frt_octbank_energies against the oracle's widget restatement: filter, y^2, exp smoothing."""
    from friture_amd.filter import IirBank
    boct, aoct = list(tabs[f"boct_{bpo}"]), list(tabs[f"aoct_{bpo}"])
    nblocks = 48 if chunk else 6
    n = 1024 * nblocks
    C = 2
    x32 = np.stack([synth("noise", n, 11), synth("chirp", n, 12)])
    alphas, kernels = dsp.band_smoothing_setup(bpo, 0.125)
    bank = IirBank(tabs["bdec"], tabs["adec"], boct, aoct, C)
    bank.set_chunk(chunk)
    got = bank.energies(x32, 1024, alphas)
    assert got.shape == (C, nblocks, 9 * bpo)
    for c in range(C if bpo == 3 else 1):
        zs = dsp.iir_bank_filtic(tabs["bdec"], tabs["adec"], boct, aoct)
        prev = [0.0] * (9 * bpo)
        for blk in range(nblocks):
            y, _, zs = dsp.iir_bank(tabs["bdec"], tabs["adec"], boct, aoct, x32[c, blk * 1024:(blk + 1) * 1024].astype(np.float64), zs)
            prev = dsp.band_energies(y, kernels, alphas, prev)
            ref = np.array(prev)
            assert np.max(np.abs(got[c, blk] / ref - 1)) <= 1e-5, (c, blk)
    # dB read-out with A weighting (octavespectrum.py:114-121)
    fi, _, _ = dsp.octave_frequencies(9 * bpo, bpo)
    A = dsp.band_weighting(fi)[0]
    bank.reset()
    db = bank.energies(x32, 1024, alphas, weight_db=A, as_db=True)
    assert np.max(np.abs(db - (10 * np.log10(got.astype(np.float64) + 1e-30) + A))) < 1e-3


@pytest.mark.parametrize("chunk,nblocks", [(512, 64), (256, 48), (512, 320)])
def test_band_energies_with_chunks_shorter_than_the_block(tabs, chunk, nblocks):
    """A time-parallel chunk shorter than the energy block (frt_octbank_set_chunk(256 / 512) with blocks of 1024): the
    block axis of the smoothing recurrence becomes the chunk and every second / fourth value is the caller's — the same
    energies as the oracle's block-by-block restatement, and the carried state serves a second call."""
    import torch
    from friture_amd.filter import IirBank
    bpo, C = 3, 2
    boct, aoct = list(tabs[f"boct_{bpo}"]), list(tabs[f"aoct_{bpo}"])
    n = 1024 * nblocks
    x32 = np.stack([synth("noise", 2 * n, 21), synth("chirp", 2 * n, 22)])
    alphas, kernels = dsp.band_smoothing_setup(bpo, 0.125)
    bank = IirBank(tabs["bdec"], tabs["adec"], boct, aoct, C)
    bank.set_chunk(chunk)
    xd = torch.from_numpy(x32).cuda()
    got = torch.cat([bank.energies(xd[:, :n].contiguous(), 1024, alphas), bank.energies(xd[:, n:].contiguous(), 1024, alphas)], dim=1).cpu().numpy()
    assert got.shape == (C, 2 * nblocks, 27)
    for c in range(C):
        zs = dsp.iir_bank_filtic(tabs["bdec"], tabs["adec"], boct, aoct)
        prev = [0.0] * 27
        for blk in range(2 * nblocks):
            y, _, zs = dsp.iir_bank(tabs["bdec"], tabs["adec"], boct, aoct, x32[c, blk * 1024:(blk + 1) * 1024].astype(np.float64), zs)
            prev = dsp.band_energies(y, kernels, alphas, prev)
            ref = np.array(prev)
            # 1e-5 per band; a band 14 decades under the block's loudest (the chirp far from it) sits at the re-association
            # noise of the time-parallel mode (1e-10 of the signal's scale) and is held to that instead
            assert np.all(np.abs(got[c, blk] - ref) <= 1e-5 * ref + 1e-14 * ref.max()), (c, blk)
    with pytest.raises(Exception):
        bank.energies(x32[:, :n], 1024, alphas)                  # host arrays: not served with a chunk below the block


@pytest.mark.parametrize("block,chunk,device", [(256, 0, False), (512, 0, False), (256, 0, True), (256, 256, True), (256, 1024, True), (512, 1024, True),
                                                (512, 512, False), (512, 256, True), (1024, 256, True), (256, 2048, False)])
def test_band_energies_with_blocks_shorter_than_1024(tabs, block, chunk, device):
    """Energy blocks of 256 / 512 samples: the two lowest-rate stages then have blocks of 1 or 2 of their samples.  Sequential
    mode, time-parallel chunks of at least a block (every entry of the block axis is the caller's: the slot kernel's short-block
    loop), and chunks shorter than the block (chunk 256 under block 512: stage 7's blocks of 2 samples ride in the lane
    kernel's groups of 4, stage 8's blocks of 1 cannot — they would span 4 entries where the caller reads every 2nd) — each
    against the oracle fed block by block."""
    import torch
    from friture_amd.filter import IirBank
    bpo, C = 3, 2
    boct, aoct = list(tabs[f"boct_{bpo}"]), list(tabs[f"aoct_{bpo}"])
    nblocks = 8192 // block * 3
    n = block * nblocks
    x32 = np.stack([synth("noise", n, 31), synth("chirp", n, 32)])
    alphas, kernels = dsp.band_smoothing_setup(bpo, 0.125)
    bank = IirBank(tabs["bdec"], tabs["adec"], boct, aoct, C)
    bank.set_chunk(chunk)
    half = n // 2 // max(block, chunk, 1) * max(block, chunk, 1)
    if device:
        xd = torch.from_numpy(x32).cuda()
        got = torch.cat([bank.energies(xd[:, :half].contiguous(), block, alphas), bank.energies(xd[:, half:].contiguous(), block, alphas)], dim=1).cpu().numpy()
    else:
        got = np.concatenate([bank.energies(x32[:, :half], block, alphas), bank.energies(x32[:, half:], block, alphas)], axis=1)
    assert got.shape == (C, nblocks, 27)
    for c in range(C):
        zs = dsp.iir_bank_filtic(tabs["bdec"], tabs["adec"], boct, aoct)
        prev = [0.0] * 27
        for blk in range(nblocks):
            y, _, zs = dsp.iir_bank(tabs["bdec"], tabs["adec"], boct, aoct, x32[c, blk * block:(blk + 1) * block].astype(np.float64), zs)
            prev = dsp.band_energies(y, kernels, alphas, prev)
            ref = np.array(prev)
            assert np.all(np.abs(got[c, blk] - ref) <= 1e-5 * ref + 1e-14 * ref.max()), (c, blk, np.max(np.abs(got[c, blk] / ref - 1)))


def test_216_band_bank_with_chunks_of_1024_and_512(tabs):
    """The 216-band bank as bench.py's configs[4] leg runs it (chunk 512: shorter than the energy block; the narrow filters' decay
    spans hundreds of chunks, so scan rows are longer than the 16 chunks whose end states stay in registers) against the chunk
    length the oracle test above uses and against the oracle itself on the first blocks."""
    import torch
    from friture_amd.filter import IirBank
    bpo, C, nblocks = 24, 2, 128
    n = 1024 * nblocks
    boct, aoct = list(tabs[f"boct_{bpo}"]), list(tabs[f"aoct_{bpo}"])
    alphas, kernels = dsp.band_smoothing_setup(bpo, 0.125)
    x32 = np.stack([synth("noise", n, 31), synth("chirp", n, 32)])
    xd = torch.from_numpy(x32).cuda()

    def run(chunk):
        b = IirBank(tabs["bdec"], tabs["adec"], boct, aoct, C)
        b.set_chunk(chunk)
        return b.energies(xd, 1024, alphas).cpu().numpy().astype(np.float64)

    ref = run(4096)
    for chunk in (1024, 512):
        e = run(chunk)
        assert e.shape == (C, nblocks, 216)
        assert np.all(np.abs(e - ref) <= 1e-5 * ref + 1e-14 * ref.max(axis=2, keepdims=True)), chunk
    e = run(512)
    zs = dsp.iir_bank_filtic(tabs["bdec"], tabs["adec"], boct, aoct)
    prev = [0.0] * 216
    for blk in range(6):
        y, _, zs = dsp.iir_bank(tabs["bdec"], tabs["adec"], boct, aoct, x32[0, blk * 1024:(blk + 1) * 1024].astype(np.float64), zs)
        prev = dsp.band_energies(y, kernels, alphas, prev)
        r = np.array(prev)
        assert np.all(np.abs(e[0, blk] - r) <= 1e-5 * r + 1e-14 * r.max()), blk



def test_energy_recurrence_split_along_time_equals_tiled(tabs):
    """Long batches run the block-energy recurrence split along time (energy_local / energy_finish kernels, >= 256 blocks),
    short calls the one-workgroup-per-channel tile kernel: a 600-block batch (ten splits, the last one ragged) must carry
    the same smoothed energies as the same samples fed in calls of 100 blocks — linear and dB read-out."""
    from friture_amd.filter import IirBank
    bpo, C, nblocks = 3, 3, 600
    boct, aoct = list(tabs[f"boct_{bpo}"]), list(tabs[f"aoct_{bpo}"])
    n = 1024 * nblocks
    x32 = np.stack([synth("noise", n, 21), synth("chirp", n, 22), synth("tone", n, 23)])
    alphas, _ = dsp.band_smoothing_setup(bpo, 0.125)
    fi, _, _ = dsp.octave_frequencies(9 * bpo, bpo)
    A = dsp.band_weighting(fi)[0]
    for as_db in (False, True):
        long_bank = IirBank(tabs["bdec"], tabs["adec"], boct, aoct, C)
        short_bank = IirBank(tabs["bdec"], tabs["adec"], boct, aoct, C)
        for b in (long_bank, short_bank):
            b.set_chunk(0)            # sequential filters: bit-identical band signals whatever the call boundaries
        kw = dict(weight_db=A, as_db=True) if as_db else {}
        whole = long_bank.energies(x32, 1024, alphas, **kw)
        parts = [short_bank.energies(x32[:, i * 102400:(i + 1) * 102400], 1024, alphas, **kw) for i in range(6)]
        pieces = np.concatenate(parts, axis=1)
        assert whole.shape == pieces.shape == (C, nblocks, 9 * bpo)
        if as_db:
            assert np.max(np.abs(whole - pieces)) < 1e-4            # float32 dB values
        else:
            assert np.max(np.abs(whole / pieces - 1)) < 1e-6         # float32 output of float64 sums
        # a second long call continues from the carried energies of the first
        again_long = long_bank.energies(x32[:, :1024 * 300], 1024, alphas, **kw)
        again_short = np.concatenate([short_bank.energies(x32[:, i * 102400:(i + 1) * 102400], 1024, alphas, **kw) for i in range(3)], axis=1)
        tol = 1e-4 if as_db else 1e-6
        err = np.max(np.abs(again_long - again_short)) if as_db else np.max(np.abs(again_long / again_short - 1))
        assert err < tol


def test_full_size_bank_properties(tabs):
    """BASELINE configs[2] size (8 ch x 2^22 samples, 1/3 octave, device resident) through size-independent
    properties: exact linearity in amplitude, streaming == batch (state and smoothed energies carried
    across calls), chunking invariance, and agreement of the first blocks with the oracle."""
    import torch
    from friture_amd.filter import IirBank
    bpo, C, n = 3, 8, 1 << 22
    boct, aoct = list(tabs[f"boct_{bpo}"]), list(tabs[f"aoct_{bpo}"])
    alphas, kernels = dsp.band_smoothing_setup(bpo, 1.0)
    gen = torch.Generator(device="cuda").manual_seed(7)
    x = 0.25 * torch.randn((C, n), generator=gen, device="cuda", dtype=torch.float32)

    def bank(chunk):
        b = IirBank(tabs["bdec"], tabs["adec"], boct, aoct, C)
        b.set_chunk(chunk)
        return b

    e = bank(2048).energies(x, 1024, alphas)
    torch.cuda.synchronize()
    assert e.shape == (C, n // 1024, 27) and bool(torch.isfinite(e).all()) and bool((e > 0).all())
    # every operation of the chain is linear or quadratic in the input and 2 is a power of two: exact
    e2 = bank(2048).energies(2.0 * x, 1024, alphas)
    assert torch.equal(e2, 4.0 * e)
    # two calls of half the length == one call (filter states and smoothed energies live in the handle)
    b = bank(2048)
    h1 = b.energies(x[:, :n // 2].contiguous(), 1024, alphas)
    h2 = b.energies(x[:, n // 2:].contiguous(), 1024, alphas)
    halves = torch.cat([h1, h2], dim=1)
    assert float(((halves - e).abs() / e).max()) < 1e-5
    # another chunking of the time axis, and the recurrence form of the zero-state pass
    for chunk in (16384, -4096):
        other = bank(chunk).energies(x, 1024, alphas)
        assert float(((other - e).abs() / e).max()) < 1e-5, chunk
    # the first 32 blocks of two channels against the oracle (sequential recurrence from zero state)
    for c in (0, C - 1):
        xs = x[c, :32 * 1024].cpu().numpy().astype(np.float64)
        zs = dsp.iir_bank_filtic(tabs["bdec"], tabs["adec"], boct, aoct)
        prev = [0.0] * 27
        for blk in range(32):
            y, _, zs = dsp.iir_bank(tabs["bdec"], tabs["adec"], boct, aoct, xs[blk * 1024:(blk + 1) * 1024], zs)
            prev = dsp.band_energies(y, kernels, alphas, prev)
            assert np.max(np.abs(e[c, blk].cpu().numpy() / np.array(prev) - 1)) <= 1e-5, (c, blk)


def test_full_size_bank_every_block_against_sequential_mode_and_oracle(tabs):
    """BASELINE configs[2] at full size, every block of every channel: the time-parallel mode as `bench.py` runs it (chunk
    1024: zero-state products on the matrix cores, row scan, one lane per chunk) against (a) the SEQUENTIAL mode of the same
    bank, which is bit-identical to the reference's recurrence (golden tests above) — all 8 x 4096 x 27 energies within 1e-5 —
    and (b) the oracle itself run over the whole of one channel (4096 blocks through oracle/iir_ref.c), within 1e-5."""
    import torch
    from friture_amd.filter import IirBank
    bpo, C, n = 3, 8, 1 << 22
    boct, aoct = list(tabs[f"boct_{bpo}"]), list(tabs[f"aoct_{bpo}"])
    alphas, kernels = dsp.band_smoothing_setup(bpo, 1.0)
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = 0.25 * torch.randn((C, n), generator=gen, device="cuda", dtype=torch.float32)
    par = IirBank(tabs["bdec"], tabs["adec"], boct, aoct, C)
    par.set_chunk(1024)
    e = par.energies(x, 1024, alphas)
    seq = IirBank(tabs["bdec"], tabs["adec"], boct, aoct, C)            # chunk 0: sequential in time
    es = seq.energies(x, 1024, alphas)
    torch.cuda.synchronize()
    assert e.shape == es.shape == (C, n // 1024, 27)
    worst = float(((e - es).abs() / es).max())
    assert worst <= 1e-5, worst
    xs = x[3].cpu().numpy().astype(np.float64)
    zs = dsp.iir_bank_filtic(tabs["bdec"], tabs["adec"], boct, aoct)
    prev = [0.0] * 27
    got = e[3].cpu().numpy()
    worst_o = 0.0
    for blk in range(n // 1024):
        y, _, zs = dsp.iir_bank(tabs["bdec"], tabs["adec"], boct, aoct, xs[blk * 1024:(blk + 1) * 1024], zs)
        prev = dsp.band_energies(y, kernels, alphas, prev)
        worst_o = max(worst_o, float(np.max(np.abs(got[blk] / np.array(prev) - 1))))
    assert worst_o <= 1e-5, worst_o


def test_configs4_bank_full_size_properties(tabs):
    """BASELINE configs[4], the filter-bank half on one GPU's shard scale: 64 ch x 2^20 samples, 1/24 octave
    (216 bands), device resident: exact linearity, chunking invariance, first blocks of two channels vs the oracle."""
    import torch
    from friture_amd.filter import IirBank
    bpo, C, n = 24, 64, 1 << 20
    boct, aoct = list(tabs[f"boct_{bpo}"]), list(tabs[f"aoct_{bpo}"])
    alphas, kernels = dsp.band_smoothing_setup(bpo, 1.0)
    gen = torch.Generator(device="cuda").manual_seed(11)
    x = 0.25 * torch.randn((C, n), generator=gen, device="cuda", dtype=torch.float32)

    def run(chunk, scale=1.0):
        b = IirBank(tabs["bdec"], tabs["adec"], boct, aoct, C)
        b.set_chunk(chunk)
        return b.energies(scale * x, 1024, alphas)

    e = run(4096)
    torch.cuda.synchronize()
    assert e.shape == (C, n // 1024, 216) and bool(torch.isfinite(e).all()) and bool((e > 0).all())
    assert torch.equal(run(4096, 2.0), 4.0 * e)
    assert float(((run(16384) - e).abs() / e).max()) < 1e-5
    for c in (0, C - 1):
        xs = x[c, :8 * 1024].cpu().numpy().astype(np.float64)
        zs = dsp.iir_bank_filtic(tabs["bdec"], tabs["adec"], boct, aoct)
        prev = [0.0] * 216
        for blk in range(8):
            y, _, zs = dsp.iir_bank(tabs["bdec"], tabs["adec"], boct, aoct, xs[blk * 1024:(blk + 1) * 1024], zs)
            prev = dsp.band_energies(y, kernels, alphas, prev)
            assert np.max(np.abs(e[c, blk].cpu().numpy() / np.array(prev) - 1)) <= 1e-5, (c, blk)


def test_randomised_streaming_bit_exact(tabs):
    """A seeded sweep of block lengths (ragged, tiny, long), channel counts and band counts through the streaming
    filter API in sequential mode: outputs, decimation factors and carried state bit-identical to the oracle."""
    from friture_amd.filter import IirBank
    rng = np.random.default_rng(7)
    for trial in range(12):
        bpo = int(rng.choice([1, 3, 6, 12, 24]))
        C = int(rng.integers(1, 4))
        boct, aoct = list(tabs[f"boct_{bpo}"]), list(tabs[f"aoct_{bpo}"])
        bank = IirBank(tabs["bdec"], tabs["adec"], boct, aoct, C)
        zs = [dsp.iir_bank_filtic(tabs["bdec"], tabs["adec"], boct, aoct) for _ in range(C)]
        for blk in range(3):
            n = int(rng.choice([1, 7, 256, 512, 1000, 1024, 4097]))
            x = 0.25 * rng.standard_normal((C, n))
            y, dec = bank.filter(x)
            for c in range(C):
                yo, deco, zs[c] = dsp.iir_bank(tabs["bdec"], tabs["adec"], boct, aoct, x[c], zs[c])
                assert list(dec) == list(deco)
                for k in range(9 * bpo):
                    assert np.array_equal(y[c][k], yo[k]), (trial, bpo, C, blk, n, c, k)


@pytest.mark.parametrize("bpo,chunk,nblocks", [(3, 1024, 96), (3, 1024, 3), (3, 2048, 40), (24, 512, 64)])
def test_look_back_output_pass_equals_the_scan_launches(tabs, option, bpo, chunk, nblocks):
    """Round 6: at the high-rate stages the output pass forms a chunk's true initial state itself — Horner over the zero-state end
    states of the K chunks the filters' decay spans, the carried state standing in front of chunk 0 — instead of waiting for a
    chunk-scan launch (csrc/iir.hip, iir_lane_body).  Same recurrence, another association order: the energies of three
    consecutive calls (carried filter states and smoothed energies; a call shorter than K chunks among the shapes) against the same
    calls with the option off — 1e-9 relative (measured ~1e-13), far inside the 1e-5 bar the oracle comparisons hold both to."""
    import torch
    from friture_amd.filter import IirBank
    C = 2
    boct, aoct = list(tabs[f"boct_{bpo}"]), list(tabs[f"aoct_{bpo}"])
    n = 1024 * nblocks
    x32 = np.stack([synth("noise", 3 * n, 31), synth("chirp", 3 * n, 32)])
    alphas, _ = dsp.band_smoothing_setup(bpo, 0.125)
    xd = torch.from_numpy(x32).cuda()
    outs = {}
    for look in (-1, 0):
        option("iir_lookback", look)
        bank = IirBank(tabs["bdec"], tabs["adec"], boct, aoct, C)
        bank.set_chunk(chunk)
        outs[look] = torch.cat([bank.energies(xd[:, i * n:(i + 1) * n].contiguous(), 1024, alphas) for i in range(3)], dim=1).cpu().numpy().astype(np.float64)
    a, b = outs[-1], outs[0]
    assert a.shape == (C, 3 * nblocks, 9 * bpo)
    assert np.all(np.abs(a - b) <= 1e-9 * np.abs(b) + 1e-14 * b.max()), float(np.max(np.abs(a - b) / (np.abs(b) + 1e-14 * b.max())))


@pytest.mark.parametrize("bpo,chunk,nblocks,C", [(3, 1024, 96, 2), (3, 512, 67, 3), (24, 512, 64, 2), (6, 256, 33, 1), (12, 1024, 16, 5), (1, 256, 9, 3)])
def test_column_form_of_the_output_pass_equals_the_lane_per_wavefront_form(tabs, option, bpo, chunk, nblocks, C):
    """Round 6: the output pass with every filter group of 64 chunks in ONE workgroup, the samples copied global -> LDS (LDS-DMA,
    swizzled rows) once per column instead of requested lane by lane by every group (csrc/iir.hip, iir_lane_col_kernel).  The
    recurrences are the same instructions on the same values: the energies of three consecutive calls (carried states; column counts
    that leave a workgroup's second column empty, a last column of fewer than 64 chunks, 1 .. 9 wavefronts per column) are BIT-IDENTICAL
    to the form it replaces, which the oracle comparisons above hold to 1e-5."""
    import torch
    from friture_amd.filter import IirBank
    boct, aoct = list(tabs[f"boct_{bpo}"]), list(tabs[f"aoct_{bpo}"])
    n = 1024 * nblocks
    x32 = np.stack([synth("noise" if c % 2 == 0 else "chirp", 3 * n, 41 + c) for c in range(C)])
    alphas, _ = dsp.band_smoothing_setup(bpo, 0.125)
    xd = torch.from_numpy(x32).cuda()
    outs = {}
    for cols in (1, 0):
        option("iir_lane_columns", cols)
        bank = IirBank(tabs["bdec"], tabs["adec"], boct, aoct, C)
        bank.set_chunk(chunk)
        outs[cols] = torch.cat([bank.energies(xd[:, i * n:(i + 1) * n].contiguous(), 1024, alphas) for i in range(3)], dim=1).cpu().numpy()
    assert outs[1].shape == (C, 3 * nblocks, 9 * bpo)
    assert np.isfinite(outs[1]).all() and outs[1].max() > 0
    assert np.array_equal(outs[1], outs[0]), float(np.max(np.abs(outs[1].astype(np.float64) - outs[0]) / (np.abs(outs[0]) + 1e-30)))
