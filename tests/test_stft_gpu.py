"""K1 parity: HIP STFT -> PSD / dB / image against the oracle and the reference's golden vectors.

Tolerances (BASELINE.json north_star: "within 1e-5 relative on the PSD"; metric defined in
SURVEY.md §8d): per frame max|gpu - ref| / max|ref| <= 1e-5 and relative L2 <= 1e-5 for the
float32 path; 1e-12 for the float64 path.  Colour words are integers: exact, except where the
normalised value sits within float32 rounding of a LUT bin edge.
"""
import numpy as np
import pytest

from conftest import rel_l2, rel_max, synth
from oracle import dsp

pytestmark = pytest.mark.gpu

TOL32 = 1e-5
TOL64 = 1e-12


def per_frame_err(gpu, ref):
    gpu, ref = np.asarray(gpu, np.float64), np.asarray(ref, np.float64)
    num = np.max(np.abs(gpu - ref), axis=-1)
    return float(np.max(num / np.max(ref, axis=-1)))


@pytest.fixture(scope="module")
def engine_cls(hip):
    from friture_amd.stft import StftEngine
    return StftEngine


@pytest.mark.parametrize("key,n_fft,hop", [("N32_hop16_noise", 32, 16), ("N256_hop64_tone", 256, 64),
                                           ("N1024_hop512_noise", 1024, 512), ("N1024_hop256_tone", 1024, 256),
                                           ("N4096_hop1024_noise", 4096, 1024), ("N16384_hop8192_tone", 16384, 8192)])
def test_psd_against_reference_golden(golden, engine_cls, key, n_fft, hop):
    g = golden("psd")
    x, ref = g[key + "_x"], g[key + "_psd"]
    got = engine_cls(n_fft, hop, 1, 32).psd(x)[0]
    assert got.shape == ref.shape
    assert per_frame_err(got, ref) <= TOL32 and rel_l2(got, ref) <= TOL32
    got64 = engine_cls(n_fft, hop, 1, 64).psd(x.astype(np.float64))[0]
    assert per_frame_err(got64, ref) <= TOL64


@pytest.mark.parametrize("n_fft", [32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384])
@pytest.mark.parametrize("overlap", [0.5, 0.75])
def test_psd_all_sizes_against_oracle(engine_cls, n_fft, overlap):
    hop = int(n_fft * (1 - overlap))
    frames = 37 if n_fft <= 2048 else 9
    T = n_fft + hop * (frames - 1) + 5
    x = np.stack([synth("noise", T, 42), synth("tone", T, 123), synth("chirp", T, 7)])
    got = engine_cls(n_fft, hop, 3, 32).psd(x)
    assert got.shape == (3, frames, n_fft // 2 + 1)
    for c in range(3):
        ref = dsp.stft_psd(x[c].astype(np.float64), n_fft, hop)
        assert per_frame_err(got[c], ref) <= TOL32, (c, per_frame_err(got[c], ref))
        assert rel_l2(got[c], ref) <= TOL32


@pytest.mark.parametrize("n_fft,hop", [(1024, 384), (1024, 513), (1024, 1), (256, 1000), (2048, 777), (16384, 16384)])
def test_psd_generic_hops(engine_cls, n_fft, hop):
    """hops that are not N/2 or N/4 (re-load path), odd hops (scalar loads), hop > N (gaps)."""
    frames = 11
    T = n_fft + hop * (frames - 1)
    x = synth("noise", T, 3)[None, :]
    got = engine_cls(n_fft, hop, 1, 32).psd(x)[0]
    ref = dsp.stft_psd(x[0].astype(np.float64), n_fft, hop)
    assert got.shape == ref.shape and per_frame_err(got, ref) <= TOL32


def test_edge_cases(engine_cls):
    e = engine_cls(1024, 512, 2, 32)
    # fewer samples than one frame: no spectra (the reference's `realizable` is 0, spectrum.py:138-142)
    assert e.psd(np.zeros((2, 1023), np.float32)).shape == (2, 0, 513)
    assert e.psd(np.zeros((2, 0), np.float32)).shape == (2, 0, 513)
    # exactly one frame, and a ragged tail that does not complete another frame
    x = np.stack([synth("noise", 1024 + 511, 1), synth("tone", 1024 + 511, 2)])
    got = e.psd(x)
    assert got.shape == (2, 1, 513)
    for c in range(2):
        assert per_frame_err(got[c], dsp.stft_psd(x[c].astype(np.float64), 1024, 512)) <= TOL32
    # silence: PSD exactly zero, dB = 10 log10(1e-30) = -300
    z = engine_cls(1024, 512, 1, 32)
    assert np.all(z.psd(np.zeros((1, 4096), np.float32)) == 0)
    assert np.allclose(z.db(np.zeros((1, 4096), np.float32)), -300.0, atol=1e-3)
    # full-scale square wave: no overflow, still within tolerance
    sq = np.sign(np.sin(np.arange(8192) * 0.3)).astype(np.float32)[None, :]
    assert per_frame_err(z.psd(sq)[0], dsp.stft_psd(sq[0].astype(np.float64), 1024, 512)) <= TOL32
    with pytest.raises(ValueError):
        e.psd(np.zeros((3, 4096), np.float32))
    from friture_amd._lib import FritureHipError
    with pytest.raises(FritureHipError):
        engine_cls(1000, 500, 1, 32)          # not a power of two
    with pytest.raises(ValueError):
        e.image(x)                            # no LUT configured


def test_run_length_invariance(engine_cls):
    """The register-shift walk must give the same spectra whatever the run length."""
    x = synth("noise", 1024 + 512 * 200, 9)[None, :]
    outs = []
    for run in (1, 3, 8, 64):
        e = engine_cls(1024, 512, 1, 32)
        e.set_run_length(run)
        outs.append(e.psd(x))
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


@pytest.mark.parametrize("frames", [1, 2, 3, 17, 64])
def test_ring_instance_of_the_headline_size(engine_cls, frames):
    """N = 1024, hop 512, channel rows on 16-byte boundaries: the PSD / dB kinds take their samples through a per-wavefront
    LDS ring filled by LDS-DMA (stft.hip, SHIFT = -1).  Every run length — a run's first two half-frames are fetched up
    front, the rest one frame ahead, the last frame fetches nothing — must give the oracle's spectra, and the same bits as
    each other (the arithmetic does not depend on where a run starts)."""
    n_fft, hop = 1024, 512
    T = n_fft + hop * (frames - 1) + 4                       # a multiple of 4: rows stay aligned for every channel
    x = np.stack([synth("noise", T, 41), synth("chirp", T, 42), synth("tone", T, 43)])
    ref = [dsp.stft_psd(x[c].astype(np.float64), n_fft, hop) for c in range(3)]
    first = None
    for run in (0, 1, 2, 3, 8, 64):
        e = engine_cls(n_fft, hop, 3, 32)
        e.set_run_length(run)
        got = e.psd(x)
        assert got.shape == (3, frames, n_fft // 2 + 1)
        for c in range(3):
            assert per_frame_err(got[c], ref[c]) <= TOL32, (run, c)
        if first is None:
            first = got
        else:
            assert np.array_equal(got, first), run
        db = e.db(x)
        strong = np.stack(ref) > 1e-6 * np.stack(ref).max(axis=2, keepdims=True)
        assert np.max(np.abs(db - 10.0 * np.log10(np.stack(ref) + 1e-30))[strong]) < 1e-3


def test_ring_and_register_window_instances_give_the_same_bits(engine_cls):
    """The colour kind keeps the register-window instance while the PSD kinds take the LDS ring: `bench.py` checks the
    image against the reference epilogue applied to the PSD of the same batch, so the two instances must agree bit for
    bit.  The window instance is reached with a device row that starts off the 16-byte grid."""
    import torch
    n_fft, hop, frames = 1024, 512, 41
    T = n_fft + hop * (frames - 1)
    host = synth("noise", T + 4, 51) + 0.3 * synth("chirp", T + 4, 52)
    buf = torch.from_numpy(host.astype(np.float32)).cuda()
    e = engine_cls(n_fft, hop, 1, 32)
    aligned = buf[:T].reshape(1, T).contiguous()                   # ring instance
    shifted = buf[1:T + 1].reshape(1, T)                           # starts 4 bytes off: register-window instance
    assert shifted.data_ptr() % 16 != 0 and aligned.data_ptr() % 16 == 0
    same = shifted.clone().contiguous()                            # the shifted samples on the grid: ring instance again
    assert same.data_ptr() % 16 == 0
    p_window = e.psd(shifted).cpu().numpy()
    p_ring = e.psd(same).cpu().numpy()
    assert np.array_equal(p_window, p_ring)
    assert np.array_equal(e.db(shifted).cpu().numpy(), e.db(same).cpu().numpy())
    assert e.psd(aligned).shape == (1, frames, n_fft // 2 + 1)


@pytest.mark.parametrize("n_fft,hop", [(4096, 1024), (8192, 4096), (16384, 8192), (16384, 4096)])
def test_lds_staged_large_frame_instance_is_reproducible(engine_cls, n_fft, hop):
    """The large-frame instances that stage the next frame in LDS wait for the copy with a hand-counted vmcnt: a long
    signal, repeated, must give the same bits every time (a race on the staging buffer would not) and the spectra of the
    instance that loads through registers (rows off the 16-byte grid) to rounding."""
    import torch
    T = 1 << 21
    gen = torch.Generator(device="cuda").manual_seed(11)
    buf = 0.25 * torch.randn(T + 4, generator=gen, device="cuda", dtype=torch.float32)
    e = engine_cls(n_fft, hop, 1, 32)
    shifted = buf[1:T + 1].reshape(1, T)
    aligned = shifted.clone().contiguous()
    assert shifted.data_ptr() % 16 != 0 and aligned.data_ptr() % 16 == 0
    ref = e.psd(shifted)
    first = e.psd(aligned).clone()
    assert float(((first - ref).abs() / ref.amax(dim=2, keepdim=True)).max()) < 2e-6
    for _ in range(6):
        assert torch.equal(e.psd(aligned), first)


@pytest.mark.parametrize("tail", [3, 8])
@pytest.mark.parametrize("n_fft", [2048, 4096, 8192, 16384])
def test_large_frame_instances(golden, engine_cls, n_fft, tail):
    """N >= 2048 has two instances of K1: the radix-16 + wave-local one (stft_big.h, the default) and the
    generic workgroup Stockham walk (selected by a negative run length).  Both must sit inside the
    tolerance of the oracle whatever the run length, and colour the same pixels up to LUT bin edges."""
    hop = n_fft // 2
    frames = 23
    # tail 3: channel rows off the 16-byte grid (register-prefetch / direct-load instances); tail 8: rows on it (the
    # instances that stage the next frame in LDS, N >= 4096)
    T = n_fft + hop * (frames - 1) + tail
    x = np.stack([synth("noise", T, 5), synth("chirp", T, 6)])
    ref = [dsp.stft_psd(x[c].astype(np.float64), n_fft, hop) for c in range(2)]
    g = golden("image")
    from friture_amd import tables
    A = tables.weighting_db(tables.rfft_frequencies(n_fft), 1e-50)[0]
    images = []
    for run in (0, 1, 5, -1, -4):
        e = engine_cls(n_fft, hop, 2, 32)
        e.set_epilogue(A, -140.0, 0.0, g["lut"])
        e.set_run_length(run)
        got = e.psd(x)
        for c in range(2):
            assert per_frame_err(got[c], ref[c]) <= TOL32, (run, c)
        img = e.image(x)
        for c in range(2):
            # the instance's pixels are exactly what the reference's float64 epilogue assigns to the instance's own PSD
            rep = dsp.image_parity(img[c], got[c], None, A, -140.0, 0.0, g["lut"])
            assert rep["epilogue_mismatch_outside_edge"] == 0 and rep["epilogue_mismatched"] <= 3, (run, c, rep)
        images.append(img)
    for im in images[1:]:
        assert np.mean(im != images[0]) < 1e-3       # different summation orders move the float32 PSD by an ulp


@pytest.mark.parametrize("n_fft,hop", [(2048, 512), (4096, 1024), (8192, 2048), (16384, 4096), (4096, 2048), (16384, 8192)])
def test_large_frame_float64_all_kinds(golden, engine_cls, n_fft, hop):
    """The float64 large-frame instance (drop-in / pitch-tracker path) through every output kind: PSD, dB, normalised and
    the colour image (4-byte pixels from an 8-byte transform: its own row pitch) against the oracle's float64 chain."""
    from friture_amd import tables
    g = golden("image")
    lut, smin, smax = g["lut"], -140.0, 0.0
    frames = 7
    T = n_fft + hop * (frames - 1) + 5
    x = np.stack([synth("noise", T, 31), synth("chirp", T, 32)]).astype(np.float64)
    A = tables.weighting_db(tables.rfft_frequencies(n_fft), 1e-50)[0]
    e = engine_cls(n_fft, hop, 2, 64)
    e.set_epilogue(A, smin, smax, lut)
    psd, db, norm, img = e.psd(x), e.db(x), e.norm(x), e.image(x)
    for c in range(2):
        ref = dsp.stft_psd(x[c], n_fft, hop)
        assert psd[c].shape == ref.shape == (frames, n_fft // 2 + 1)
        assert per_frame_err(psd[c], ref) <= TOL64
        ref_db = 10.0 * np.log10(ref + 1e-30) + A
        # bins 60 dB and more below the frame maximum carry the transform's absolute error (1e-16 of the maximum) as a
        # large relative one: dB and colour of those are ill-conditioned in any arithmetic
        strong = ref > 1e-6 * ref.max(axis=1, keepdims=True)
        assert np.max(np.abs(db[c] - ref_db)[strong]) < 1e-8
        assert np.max(np.abs(norm[c] - (ref_db - smin) / (smax - smin))[strong]) < 1e-10
        idx, frac = dsp.colour_index(ref, A, smin, smax)
        differs = img[c] != lut[idx]
        assert np.mean(differs[strong]) < 1e-4, (c, np.mean(differs[strong]))      # only pixels at an index edge may differ
        assert np.mean(differs) < 2e-2, (c, np.mean(differs))


def test_db_norm_image_against_golden(golden, engine_cls):
    g = golden("image")
    x, A, lut = g["x"], g["weight"], g["lut"]
    smin, smax = float(g["spec_min"]), float(g["spec_max"])
    e = engine_cls(1024, 512, 1, 32)
    e.set_epilogue(A, smin, smax, lut)
    ref_norm = g["norm"].T                       # reference is (bins, frames)
    ref_db = ref_norm * (smax - smin) + smin
    db = e.db(x)[0]
    norm = e.norm(x)[0]
    psd_ref = dsp.stft_psd(x.astype(np.float64), 1024, 512)
    strong = psd_ref > 1e-6 * psd_ref.max(axis=1, keepdims=True)   # dB of near-cancelled bins is ill-conditioned in f32
    assert np.max(np.abs(db - ref_db)[strong]) < 1e-3
    assert np.max(np.abs(norm - ref_norm)[strong]) < 1e-5
    # bins -60 dB and more below the frame maximum: float32 PSD error (1e-7 of the maximum) dominates
    assert np.max(np.abs(db - ref_db)) < 5.0
    # Colour image (integer output).  Two separate statements (pixel-exact except where v*255 is within
    # 1e-6 of an integer):
    #  (1) the epilogue (dB -> weighting -> normalise -> index -> LUT) is EXACT given the float32 power spectrum the
    #      same path produces: the reference's float64 operations applied to the GPU's PSD give the GPU's pixels;
    #  (2) against the reference's own image the only differing pixels are those whose float32 power (1e-7 of the
    #      frame maximum off) lands on the other side of an index edge — counted, and bounded by the PSD tolerance.
    img = e.image(x)[0]
    ref_img = g["image"].T
    rep = dsp.image_parity(img, e.psd(x)[0], psd_ref, A, smin, smax, lut)
    assert rep["epilogue_mismatch_outside_edge"] == 0, rep
    assert rep["epilogue_mismatched"] <= 2, rep                       # pixels within 1e-6 of an edge: ~1e-6 of all
    assert rep["psd_rel_max"] <= TOL32
    assert np.array_equal(img != ref_img, img != lut[dsp.colour_index(psd_ref, A, smin, smax)[0]])
    # against the float64 reference image: every differing pixel must be a bin that the float32 PSD's measured error
    # (at most psd_rel_max of the frame maximum per bin) can carry across the edge it crossed — accounted one by one
    assert rep["mismatch_unaccounted"] == 0, rep
    assert rep["psd_rel_max"] <= 1e-5, rep
    assert np.mean((img != ref_img)[strong]) < 2e-4, rep          # bins above the float32 error floor: a handful at edges
    # float64 instance: pixel-exact everywhere but at exact bin edges
    e64 = engine_cls(1024, 512, 1, 64)
    e64.set_epilogue(A, smin, smax, lut)
    img64 = e64.image(x.astype(np.float64))[0]
    assert np.mean(img64 != ref_img) < 1e-4
    assert np.max(np.abs(e64.norm(x.astype(np.float64))[0] - ref_norm)) < 1e-9


@pytest.mark.parametrize("n_fft,hop,frames", [(1024, 512, 300), (1024, 256, 120), (512, 256, 200), (64, 32, 300),
                                               (1024, 333, 60), (2048, 1024, 60), (2048, 777, 20), (4096, 1024, 40),
                                               (8192, 4096, 24), (16384, 8192, 20)])
@pytest.mark.parametrize("weighting,smin,smax", [("A", -140.0, 0.0), (None, -100.0, -20.0), ("C", -63.7, -3.3)])
def test_image_epilogue_exact(golden, engine_cls, n_fft, hop, frames, weighting, smin, smax):
    """The float32 IMAGE kind colours exactly the pixels the reference's float64 epilogue assigns to the float32 PSD
    of the same path, for every kernel instance (one-wave frames, register-shift hops, generic walk, large frames),
    with and without weighting, for wide and narrow dB ranges — no exemption beyond 1e-6 of an index edge."""
    from friture_amd import tables
    lut = golden("image")["lut"]
    T = n_fft + hop * (frames - 1) + 2
    x = np.stack([synth("noise", T, 21), synth("tone", T, 22), 1e-3 * synth("chirp", T, 23)])
    w = None if weighting is None else tables.weighting_db(tables.rfft_frequencies(n_fft), 1e-50)["ABC".index(weighting)]
    e = engine_cls(n_fft, hop, 3, 32)
    e.set_epilogue(w, smin, smax, lut)
    psd, img = e.psd(x), e.image(x)
    for c in range(3):
        rep = dsp.image_parity(img[c], psd[c], None, w, smin, smax, lut)
        assert rep["epilogue_mismatch_outside_edge"] == 0, (c, rep)
        assert rep["epilogue_mismatched"] <= 3, (c, rep)
    # digital silence and a full-scale square wave: floors, ceilings and exact powers of two
    z = np.zeros((3, T), np.float32)
    z[1] = np.where(np.arange(T) % 64 < 32, 1.0, -1.0)
    z[2, ::n_fft // 4] = 1.0
    psd, img = e.psd(z), e.image(z)
    for c in range(3):
        rep = dsp.image_parity(img[c], psd[c], None, w, smin, smax, lut)
        assert rep["epilogue_mismatch_outside_edge"] == 0, (c, rep)


def test_audioproc_dropin(golden, hip):
    """friture_amd.audioproc.audioproc against reference outputs of friture.audioproc.audioproc."""
    from friture_amd.audioproc import audioproc
    g = golden("psd")
    p = audioproc()
    for key, n_fft, hop in [("N1024_hop512_noise", 1024, 512), ("N32_hop16_noise", 32, 16),
                            ("N16384_hop8192_tone", 16384, 8192)]:
        p.set_fftsize(n_fft)
        x = g[key + "_x"].astype(np.float64)
        for f in range(2):
            got = p.analyzelive(x[f * hop:f * hop + n_fft])
            assert got.dtype == np.float64 and rel_max(got, g[key + "_psd"][f]) <= TOL64
        assert np.array_equal(p.window, dsp.hann_symmetric(n_fft))
    p.set_fftsize(1024)
    assert np.array_equal(p.get_freq_scale(), g["N1024_freq"])
    for got, want in zip(p.get_freq_weighting(), (g["N1024_A"], g["N1024_B"], g["N1024_C"])):
        assert np.array_equal(got, want)
    assert p.size_sq == 1024.0 ** 2
    spec = np.fft.rfft(np.ones(1024))
    assert np.array_equal(p.norm_square(spec), (spec * spec.conjugate()).real / 1024.0 ** 2)
    with pytest.raises(ValueError):
        p.analyzelive(np.zeros(1000))


def test_device_resident_full_size_properties(engine_cls):
    """BASELINE configs[1] size (1 ch, T = 2^26, N = 1024, hop 512 -> 131 071 spectra) on torch
    CUDA tensors, checked through size-independent properties: Parseval per frame, agreement of a
    random sample of frames with the oracle, and linearity in amplitude."""
    import torch
    T = 1 << 26
    gen = torch.Generator(device="cuda").manual_seed(42)
    x = 0.25 * torch.randn((1, T), generator=gen, device="cuda", dtype=torch.float32)
    e = engine_cls(1024, 512, 1, 32)
    psd = e.psd(x)
    torch.cuda.synchronize()
    F = e.frames_for(T)
    assert psd.shape == (1, F, 513) and F == 131071
    # Parseval: sum_n (x w)^2 = (P0 + 2 sum_{0<k<N/2} Pk + P_{N/2}) * N   (P = |X|^2 / N^2)
    w = torch.tensor(dsp.hann_symmetric(1024), device="cuda")
    frames = x[0].unfold(0, 1024, 512).double() * w
    lhs = (frames ** 2).sum(dim=1)
    p = psd[0].double()
    rhs = (p[:, 0] + 2 * p[:, 1:512].sum(dim=1) + p[:, 512]) * 1024.0
    assert float(((lhs - rhs).abs() / lhs).max()) < 1e-5
    # sampled frames against the oracle
    xs = x[0].cpu().numpy()
    rng = np.random.default_rng(0)
    for f in list(rng.integers(0, F, 64)) + [0, F - 1]:
        ref = dsp.psd_frame(xs[f * 512:f * 512 + 1024].astype(np.float64), dsp.hann_symmetric(1024))
        assert rel_max(psd[0, f].cpu().numpy(), ref) <= TOL32
    # linearity: PSD(2x) = 4 PSD(x) exactly in binary floating point
    psd2 = e.psd(2.0 * x)
    assert torch.equal(psd2, 4.0 * psd)


def test_full_size_headline_every_frame_against_the_float64_instance(engine_cls):
    """BASELINE configs[1] at full size, every one of the 131 071 spectra and 67 M pixels: the float32 instance that `bench.py`
    times against the float64 instance of the same kernel — which agrees with the reference to 1e-12 (golden tests above) —
    (a) PSD within 1e-5 of every frame's maximum; (b) the colour image: the reference's float64 epilogue (spectrogram.py:
    119-129, lookup_table.py:50-52) is evaluated on the float64 PSD with torch on the device, and EVERY differing pixel must be
    a bin that the measured float32 PSD error of its frame can carry across the index edge it crossed (the accounting of
    oracle/dsp.py:image_parity, here over the whole batch instead of its first 4096 frames)."""
    import torch
    from friture_amd import palette, tables
    T, n_fft, hop = 1 << 26, 1024, 512
    x = torch.from_numpy((0.25 * np.random.default_rng(42).standard_normal(T, dtype=np.float32))[None]).cuda()
    weight = tables.weighting_db(tables.rfft_frequencies(n_fft), 1e-50)[0]
    lut = palette.cmr_lut()
    smin, smax = -140.0, 0.0
    e32, e64 = engine_cls(n_fft, hop, 1, 32), engine_cls(n_fft, hop, 1, 64)
    e32.set_epilogue(weight, smin, smax, lut)
    p32 = e32.psd(x)[0]
    img = e32.image(x)[0]
    p64 = e64.psd(x.double())[0]
    torch.cuda.synchronize()
    assert p32.shape == p64.shape == img.shape == (131071, 513)
    err = (p32.double() - p64).abs().amax(dim=1, keepdim=True)                  # e_f: every bin's |dP| <= e_f
    assert float((err[:, 0] / p64.amax(dim=1)).max()) <= TOL32
    w = torch.from_numpy(np.asarray(weight, np.float64)).cuda()
    q = 255.0 * ((10.0 * torch.log10(p64 + 1e-30) + w[None, :] - smin) / (smax - smin))
    idx = (q.clamp(0.0, 255.0)).to(torch.int64)                                  # int(clip(v, 0, 1) * 255): truncation
    ref_img = torch.from_numpy(lut.view(np.int32)).cuda()[idx]
    bad = img != ref_img
    n_bad = int(bad.sum())
    assert n_bad <= 1e-5 * img.numel(), n_bad
    if n_bad:
        g = 2550.0 / (np.log(10.0) * abs(smax - smin))
        rows, cols = torch.nonzero(bad, as_tuple=True)
        pb, eb, qb = (p64[rows, cols] + 1e-30).cpu().numpy(), err[rows, 0].cpu().numpy(), q[rows, cols].cpu().numpy()
        with np.errstate(divide="ignore", invalid="ignore"):
            up = g * np.log1p(eb / pb)
            down = np.where(eb < pb, -g * np.log1p(-np.minimum(eb / pb, 1.0 - 1e-16)), np.inf)
        reach = np.maximum(up, down) + 1e-9
        to_edge = np.abs(qb - np.clip(np.rint(qb), 1.0, 255.0))
        assert np.all(to_edge <= reach), (n_bad, float(np.max(to_edge - reach)))


def test_large_frame_shard_full_size_properties(engine_cls):
    """BASELINE configs[3], one GPU's shard (32 of 256 ch, T = 2^20, N = 16384, hop 8192): Parseval per frame,
    sampled frames against the oracle, exact linearity in amplitude."""
    import torch
    C, T, N, hop = 32, 1 << 20, 16384, 8192
    gen = torch.Generator(device="cuda").manual_seed(43)
    x = 0.25 * torch.randn((C, T), generator=gen, device="cuda", dtype=torch.float32)
    e = engine_cls(N, hop, C, 32)
    psd = e.psd(x)
    torch.cuda.synchronize()
    F = e.frames_for(T)
    assert psd.shape == (C, F, N // 2 + 1) and F == 127
    w = torch.tensor(dsp.hann_symmetric(N), device="cuda")
    for c in (0, 17, 31):
        frames = x[c].unfold(0, N, hop).double() * w
        lhs = (frames ** 2).sum(dim=1)
        p = psd[c].double()
        rhs = (p[:, 0] + 2 * p[:, 1:N // 2].sum(dim=1) + p[:, N // 2]) * N
        assert float(((lhs - rhs).abs() / lhs).max()) < 1e-5
    rng = np.random.default_rng(1)
    win = dsp.hann_symmetric(N)
    for c, f in zip(rng.integers(0, C, 12), rng.integers(0, F, 12)):
        xs = x[c, f * hop:f * hop + N].cpu().numpy().astype(np.float64)
        assert rel_max(psd[c, f].cpu().numpy(), dsp.psd_frame(xs, win)) <= TOL32
    assert torch.equal(e.psd(2.0 * x), 4.0 * psd)


def test_randomised_shapes_against_oracle(engine_cls):
    """A seeded sweep over frame size, hop (aligned, odd, larger than the frame), channel count, frame count,
    precision and run length: every kernel instance and dispatch branch against the oracle."""
    rng = np.random.default_rng(20260924)
    sizes = [32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384]
    for trial in range(120):
        n_fft = int(rng.choice(sizes))
        mode = int(rng.integers(0, 4))
        hop = {0: n_fft // 2, 1: n_fft // 4, 2: int(rng.integers(1, 2 * n_fft)), 3: 2 * int(rng.integers(1, n_fft))}[mode]
        C = int(rng.integers(1, 5))
        frames = int(rng.integers(1, 40 if n_fft <= 2048 else 12))
        T = n_fft + hop * (frames - 1) + int(rng.integers(0, hop))
        precision = 32 if rng.random() < 0.75 else 64
        x = (0.25 * rng.standard_normal((C, T))).astype(np.float32 if precision == 32 else np.float64)
        e = engine_cls(n_fft, hop, C, precision)
        run = int(rng.choice([0, 0, 1, 3, 8, -2]))
        e.set_run_length(run)
        got = e.psd(x)
        assert got.shape == (C, frames, n_fft // 2 + 1), (trial, n_fft, hop, C, frames)
        tol = TOL32 if precision == 32 else TOL64
        for c in range(C):
            ref = dsp.stft_psd(x[c].astype(np.float64), n_fft, hop)
            assert per_frame_err(got[c], ref) <= tol, (trial, n_fft, hop, C, frames, precision, run, c)


# ---- split output rows (frt_stft_run_split): rows of N/2 values on 64-byte boundaries + a Nyquist plane -------------------

def _packed_from_split(rows, nyq):
    return np.concatenate([np.asarray(rows), np.asarray(nyq)[..., None]], axis=-1)


@pytest.mark.parametrize("n_fft", [32, 64, 128, 256, 512, 1024])
@pytest.mark.parametrize("precision", [32, 64])
def test_split_rows_equal_packed_rows_bit_for_bit(engine_cls, n_fft, precision):
    """Every N <= 1024 instance (hop N/2, N/4: register window / ring; generic even and odd hops: reload path), every output
    kind, host buffers and device tensors, several run lengths: the split layout holds the packed layout's values at other
    addresses — and the packed layout is what the oracle / golden tests above pin."""
    import torch
    from friture_amd import palette, tables
    weight = tables.weighting_db(tables.rfft_frequencies(n_fft), 1e-50)[0]
    lut = palette.cmr_lut()
    dt = np.float32 if precision == 32 else np.float64
    for hop in (n_fft // 2, n_fft // 4, 3 * n_fft // 8, n_fft // 2 + 1):
        for C, frames, run in ((1, 1, 0), (2, 67, 0), (3, 130, 64), (1, 23, 3), (2, 200, 100)):
            T = n_fft + hop * (frames - 1) + 4
            x = np.stack([synth(("noise", "tone", "chirp")[c % 3], T, 11 + c) for c in range(C)]).astype(dt)
            e = engine_cls(n_fft, hop, C, precision)
            e.set_epilogue(weight, -140.0, 0.0, lut)
            e.set_run_length(run)
            xd = torch.from_numpy(x).cuda()
            for kind in (0, 1, 2, 3):
                packed = e.run(kind, x)
                rows, nyq = e.run_split(kind, x)
                assert rows.shape == (C, frames, n_fft // 2) and nyq.shape == (C, frames)
                assert np.array_equal(_packed_from_split(rows, nyq).view(np.uint8), packed.view(np.uint8)), (hop, C, frames, run, kind, "host")
                drows, dnyq = e.run_split(kind, xd)
                torch.cuda.synchronize()
                got = _packed_from_split(drows.cpu().numpy(), dnyq.cpu().numpy())
                assert np.array_equal(got.view(np.uint8), packed.view(np.uint8)), (hop, C, frames, run, kind, "device")


def test_split_rows_against_oracle_and_golden(golden, engine_cls):
    """The split layout against the reference-recorded golden PSD and the oracle directly (not only through the packed rows)."""
    g = golden("psd")
    for key, n_fft, hop in (("N1024_hop512_noise", 1024, 512), ("N1024_hop256_tone", 1024, 256), ("N256_hop64_tone", 256, 64)):
        x, ref = g[key + "_x"], g[key + "_psd"]
        rows, nyq = engine_cls(n_fft, hop, 1, 32).run_split(0, x)
        assert per_frame_err(_packed_from_split(rows, nyq)[0], ref) <= TOL32
        rows, nyq = engine_cls(n_fft, hop, 1, 64).run_split(0, x.astype(np.float64))
        assert per_frame_err(_packed_from_split(rows, nyq)[0], ref) <= TOL64
    x = synth("noise", 1024 + 512 * 99, 5)
    rows, nyq = engine_cls(1024, 512, 1, 32).run_split(0, x)
    assert per_frame_err(_packed_from_split(rows, nyq)[0], dsp.stft_psd(x.astype(np.float64), 1024, 512)) <= TOL32


def test_split_rows_edge_cases(engine_cls):
    from friture_amd._lib import FritureHipError
    e = engine_cls(1024, 512, 2, 32)
    rows, nyq = e.run_split(0, np.zeros((2, 1023), np.float32))            # fewer samples than a frame: no spectra
    assert rows.shape == (2, 0, 512) and nyq.shape == (2, 0)
    with pytest.raises(FritureHipError) as err:
        engine_cls(2048, 1024, 1, 32).run_split(0, np.zeros((1, 8192), np.float32))
    assert err.value.status == -4                                          # FRT_ERR_UNSUPPORTED: split rows are an N <= 1024 layout


def test_split_rows_full_size_headline_image(engine_cls):
    """BASELINE configs[1] at full size in the layout bench.py times (split rows, colour kind): every one of the 131 071 x 513
    pixels equals the packed-layout image, whose every pixel is accounted for against the float64 reference by
    test_full_size_headline_every_frame_against_the_float64_instance; the rows lie on 64-byte boundaries."""
    import torch
    from friture_amd import palette, tables
    T, n_fft, hop = 1 << 26, 1024, 512
    x = torch.from_numpy((0.25 * np.random.default_rng(42).standard_normal(T, dtype=np.float32))[None]).cuda()
    weight = tables.weighting_db(tables.rfft_frequencies(n_fft), 1e-50)[0]
    e = engine_cls(n_fft, hop, 1, 32)
    e.set_epilogue(weight, -140.0, 0.0, palette.cmr_lut())
    for kind in (3, 0):
        packed = e.run(kind, x)
        rows, nyq = e.run_split(kind, x)
        torch.cuda.synchronize()
        assert rows.shape == (1, 131071, 512) and rows.data_ptr() % 64 == 0 and rows.stride(1) * rows.element_size() % 64 == 0
        assert torch.equal(rows.view(torch.int32), packed[..., :512].view(torch.int32))
        assert torch.equal(nyq.view(torch.int32), packed[..., 512].view(torch.int32))


def test_prepared_launches_equal_run_and_run_split(hip):
    """StftEngine.prepare: a repeated launch whose shapes, pointers and stream were taken once — the callable is one
    frt_stft_run / frt_stft_run_split call.  Bit-identical to run() / run_split() on the same buffers, both layouts, two kinds; wrong
    shapes are refused at prepare time."""
    import torch
    from friture_amd import palette, tables
    from friture_amd.stft import StftEngine
    N, hop, C, T = 1024, 512, 2, 1024 + 512 * 99
    rng = np.random.default_rng(3)
    x = torch.from_numpy((0.25 * rng.standard_normal((C, T))).astype(np.float32)).cuda()
    eng = StftEngine(N, hop, C, 32)
    eng.set_epilogue(tables.weighting_db(tables.rfft_frequencies(N), 1e-50)[0], -140.0, 0.0, palette.cmr_lut())
    F = eng.frames_for(T)
    for kind, dt in ((0, torch.float32), (3, torch.int32)):
        want = eng.run(kind, x)
        out = torch.zeros((C, F, N // 2 + 1), dtype=dt, device="cuda")
        go = eng.prepare(kind, x, out)
        go()
        go()
        assert torch.equal(out, want)
        rows, nyq = torch.zeros((C, F, N // 2), dtype=dt, device="cuda"), torch.zeros((C, F), dtype=dt, device="cuda")
        go = eng.prepare(kind, x, rows, nyq)
        go()
        assert torch.equal(rows, want[:, :, :N // 2]) and torch.equal(nyq, want[:, :, N // 2])
    with pytest.raises(ValueError):
        eng.prepare(0, x, torch.zeros((C, F + 1, N // 2 + 1), dtype=torch.float32, device="cuda"))
    with pytest.raises(ValueError):
        eng.prepare(3, x, torch.zeros((C, F, N // 2 + 1), dtype=torch.float32, device="cuda"))
