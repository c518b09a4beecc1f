"""oracle/make_golden.py — record golden vectors from the unmodified reference and pin the oracle.

Run in the build container (needs the reference checkout, see oracle/refshim.py):

    python -m oracle.make_golden

For every row of the hot-path table this script
  1. runs the *reference's own code* on seeded inputs,
  2. checks that oracle/dsp.py reproduces it (bit-exactly unless noted), and
  3. writes the reference outputs to tests/golden/*.npz so that the same comparison can be replayed
     where the reference checkout does not exist (tests/test_oracle_golden.py) and so that the
     HIP backend can be compared with reference outputs directly (tests/test_*_gpu.py).
"""
from __future__ import annotations

import hashlib
import sys
from pathlib import Path

import numpy as np

from . import dsp, refshim

GOLD = Path(__file__).resolve().parents[1] / "tests" / "golden"


def noise(seed, n, scale=0.25):
    return scale * np.random.default_rng(seed).standard_normal(n)


def tone(seed, n, f=1000.0):
    t = np.arange(n)
    return 0.5 * np.sin(2 * np.pi * f * t / 48000.0) + 1e-3 * np.random.default_rng(seed).standard_normal(n)


def as_f32_f64(x):
    """audio is float32 promoted to float64 (friture/audiobackend.py:466-468)."""
    return x.astype(np.float32).astype(np.float64)


def same(name, a, b, tol=0.0):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    if tol == 0.0:
        ok = np.array_equal(a, b)
        err = 0.0 if ok else float(np.max(np.abs(a.astype(float) - b.astype(float))))
    else:
        err = float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
        ok = err <= tol
    print(f"  {'ok ' if ok else 'BAD'} oracle vs reference: {name} (err {err:.3e}, tol {tol:g})")
    if not ok:
        raise SystemExit(f"oracle does not reproduce the reference for {name}")


def main():
    refshim.install()
    from friture.audioproc import audioproc
    from friture.filter import (octave_filter_bank_decimation, octave_filter_bank_decimation_filtic,
                                octave_frequencies)
    from friture.octavefilters import Octave_Filters
    from friture.ringbuffer import RingBuffer
    from friture.signal.color_tranform import Color_Transform
    from friture.signal.correlation import generalized_cross_correlation
    from friture.signal.decimate import decimate_multiple, decimate_multiple_filtic
    from friture.signal.exp_smoothing import exp_smoothed_value, exp_smoothed_value_2d
    from friture.signal.frequency_resampler import Frequency_Resampler
    from friture.signal.lfilter import lfilter_float64_1D
    from friture.signal.online_linear_2D_resampler import Online_Linear_2D_resampler
    import friture.plotting.frequency_scales as fscales
    from friture import generated_filters

    GOLD.mkdir(parents=True, exist_ok=True)

    # ---- P1/P2/P3: PSD ---------------------------------------------------------------------------
    print("P1-P3 audioproc.analyzelive / STFT loop")
    out = {}
    for n_fft, hop, frames, sig in [(32, 16, 6, "noise"), (256, 64, 5, "tone"), (1024, 512, 6, "noise"),
                                    (1024, 256, 5, "tone"), (4096, 1024, 3, "noise"), (16384, 8192, 2, "tone")]:
        T = n_fft + hop * (frames - 1)
        x = as_f32_f64(noise(42, T) if sig == "noise" else tone(123, T))
        proc = audioproc()
        proc.set_fftsize(n_fft)
        ref = np.stack([proc.analyzelive(x[f * hop:f * hop + n_fft]) for f in range(frames)])
        same(f"psd N={n_fft} hop={hop} {sig}", dsp.stft_psd(x, n_fft, hop), ref)
        key = f"N{n_fft}_hop{hop}_{sig}"
        out[key + "_x"] = x.astype(np.float32)
        out[key + "_psd"] = ref
        if n_fft == 1024 and hop == 512:
            same("window", dsp.hann_symmetric(n_fft), proc.window)
            same("freq", dsp.frequency_axis(n_fft), proc.get_freq_scale())
            for nm, o, r in zip("ABC", dsp.weighting_curves(proc.freq), proc.get_freq_weighting()):
                same(f"weighting {nm}", o, r)
            out["N1024_A"], out["N1024_B"], out["N1024_C"] = proc.get_freq_weighting()
            out["N1024_freq"] = proc.get_freq_scale()
    np.savez_compressed(GOLD / "psd.npz", **out)

    # ---- P4/P7: dB + weighting + normalise + colour ----------------------------------------------
    print("P4/P7 spectrogram image")
    n_fft, hop, frames = 1024, 512, 12
    x = as_f32_f64(tone(123, n_fft + hop * (frames - 1)) + noise(42, n_fft + hop * (frames - 1), 0.05))
    proc = audioproc()
    proc.set_fftsize(n_fft)
    spn = np.stack([proc.analyzelive(x[f * hop:f * hop + n_fft]) for f in range(frames)], axis=1)  # (bins, frames)
    A = proc.get_freq_weighting()[0]
    spec_min, spec_max = -140.0, 0.0
    norm = (10.0 * np.log10(spn + 1e-30) + A[:, None] - spec_min) / (spec_max - spec_min)   # spectrogram.py:119-129,161-162
    ct = Color_Transform()
    img = ct.push(norm)
    lut = dsp.colour_lut(dsp.cmrmap())
    same("colour LUT", lut, ct.colors)
    same("image", dsp.spectrogram_image(x, n_fft, hop, A, spec_min, spec_max, lut).T, img)
    np.savez_compressed(GOLD / "image.npz", x=x.astype(np.float32), weight=A, spec_min=spec_min, spec_max=spec_max,
                        lut=ct.colors, norm=norm, image=img)

    # ---- P5/P6: screen-space resamplers ----------------------------------------------------------
    print("P5/P6 resamplers")
    res = {}
    freq = proc.get_freq_scale()
    for scale_name, scale in [("linear", fscales.Linear), ("log", fscales.Logarithmic), ("mel", fscales.Mel),
                              ("erb", fscales.Erb), ("octave", fscales.Octave)]:
        fr = Frequency_Resampler(scale, 20.0, 20000.0, 100)
        fr.setfreq(freq)
        r = fr.push(norm)
        tg = dsp.frequency_targets(scale_name, 20.0, 20000.0, 100)
        same(f"freq targets {scale_name}", tg, fr.xscaled)
        same(f"freq resample {scale_name}", dsp.frequency_resample(tg, freq, norm), r)
        res[f"fr_{scale_name}"] = r
        res[f"fr_{scale_name}_targets"] = fr.xscaled
    # time resampler: STFT rate 93.75 cols/s -> 60 px/s and an up-sampling case, fed in two pushes
    for tag, (L, M) in {"down": (25, 16), "up": (3, 7)}.items():
        tr = Online_Linear_2D_resampler(L, M, 100)
        mine = dsp.TimeResampler(L, M, 100)
        a = tr.push(res["fr_mel"][:, :5])
        b = tr.push(res["fr_mel"][:, 5:])
        same(f"time resample {tag} 1", mine.push(res["fr_mel"][:, :5]), a)
        same(f"time resample {tag} 2", mine.push(res["fr_mel"][:, 5:]), b)
        res[f"tr_{tag}_a"], res[f"tr_{tag}_b"] = a, b
    # a resize between pushes: 100 -> 137 -> 64 rows (set_height Fourier-resamples the carried column, scipy_resample.py)
    from friture.signal.scipy_resample import resample as ref_resample
    for h_new in (137, 64, 100, 211):
        col = res["fr_mel"][:, 7]
        same(f"fourier resample 100 -> {h_new}", dsp.fourier_resample(col, h_new), ref_resample(col, h_new), tol=1e-13)
        res[f"fourier_100_{h_new}"] = ref_resample(col, h_new)
    tr = Online_Linear_2D_resampler(25, 16, 100)
    mine = dsp.TimeResampler(25, 16, 100)
    fr137 = Frequency_Resampler(fscales.Mel, 20.0, 20000.0, 137)
    fr137.setfreq(freq)
    fr64 = Frequency_Resampler(fscales.Mel, 20.0, 20000.0, 64)
    fr64.setfreq(freq)
    ncol = norm.shape[1]
    c1, c2 = ncol // 3, 2 * ncol // 3
    seq = [res["fr_mel"][:, :c1], fr137.push(norm)[:, c1:c2], fr64.push(norm)[:, c2:]]
    for i, block in enumerate(seq):
        a = tr.push(block)
        same(f"time resample across resize {i}", mine.push(block), a, tol=0.0 if i == 0 else 1e-13)
        res[f"tr_resize_in_{i}"], res[f"tr_resize_out_{i}"] = block, a
    res["norm"], res["freq"] = norm, freq
    np.savez_compressed(GOLD / "pipeline.npz", **res)

    # ---- P8: exponential smoothing ----------------------------------------------------------------
    print("P8 exp smoothing")
    rng = np.random.default_rng(7)
    kern = dsp.smoothing_kernel(0.02, 64)
    d1 = rng.standard_normal(40) ** 2
    d2 = rng.standard_normal((5, 100)) ** 2
    prev = rng.standard_normal(5) ** 2
    r1 = exp_smoothed_value(kern, 0.02, d1, 0.3)
    r2 = exp_smoothed_value_2d(kern, 0.02, d2, prev)
    r3 = exp_smoothed_value_2d(kern, 0.02, d2[:, :17], prev)
    same("exp 1d", dsp.exp_smoothed_value(kern, 0.02, d1, 0.3), r1)
    same("exp 2d long", dsp.exp_smoothed_value_2d(kern, 0.02, d2, prev), r2)
    same("exp 2d short", dsp.exp_smoothed_value_2d(kern, 0.02, d2[:, :17], prev), r3)
    np.savez_compressed(GOLD / "exp_smoothing.npz", kern=kern, d1=d1, d2=d2, prev=prev, r1=r1, r2=r2, r3=r3)

    # ---- spectrum widget post-processing (smoothing, dB, peak, harmonic product spectrum) -------------
    # Spectrum_Widget cannot be imported (QObject base); its pure method harmonic_product_spectrum is
    # lifted out of the unmodified source text and executed as is.
    print("spectrum widget read-out")
    import ast
    src = (Path(refshim.REFERENCE_ROOT) / "friture" / "spectrum.py").read_text()
    fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "harmonic_product_spectrum")
    ns = {"ones": np.ones}
    exec(compile(ast.Module([fn], []), "friture/spectrum.py", "exec"), ns)
    n_fft, hop, frames = 1024, 256, 9
    xs = as_f32_f64(tone(123, n_fft + hop * (frames - 1), 440.0) + tone(5, n_fft + hop * (frames - 1), 880.0))
    proc = audioproc()
    proc.set_fftsize(n_fft)
    spn = np.stack([proc.analyzelive(xs[f * hop:f * hop + n_fft]) for f in range(frames)], axis=1)
    alpha = 1.0 - (1.0 - 0.65) ** (1.0 / (0.025 * 48000 / hop + 1))        # spectrum.py:196-222, 25 ms response
    kern = (1.0 - alpha) ** np.arange(2 * 4096 - 1, -1, -1)
    prev = np.zeros(513)
    sp_ref = exp_smoothed_value_2d(kern, alpha, spn, prev)
    hps_ref = ns["harmonic_product_spectrum"](None, sp_ref)
    same("harmonic product spectrum", dsp.harmonic_product_spectrum(sp_ref), hps_ref)
    wA = proc.get_freq_weighting()[0]
    ro = dsp.spectrum_readout(spn, kern, alpha, prev, wA, proc.get_freq_scale())
    db_ref = 10.0 * np.log10(sp_ref + 1e-30) + wA
    same("spectrum dB", ro["db"], db_ref)
    assert ro["peak_index"] == int(np.argmax(db_ref)) and ro["pitch_index"] == int(np.argmax(hps_ref))
    np.savez_compressed(GOLD / "spectrum.npz", x=xs.astype(np.float32), spn=spn, kern_alpha=alpha, smoothed=sp_ref, db=db_ref,
                        hps=hps_ref, peak_index=int(np.argmax(db_ref)), pitch_index=int(np.argmax(hps_ref)), weight=wA)

    # ---- O1/G2: IIR ------------------------------------------------------------------------------
    print("O1/G2 lfilter, decimate, exact IIR bank")
    bdec, adec = [np.array(v) for v in generated_filters.PARAMS["dec"]]
    tabs = dsp.load_filter_tables()
    same("bdec table", tabs["bdec"], bdec)
    same("adec table", tabs["adec"], adec)
    x = as_f32_f64(noise(42, 4 * 512))
    y_ref, z_ref = lfilter_float64_1D(bdec, adec, x[:700], np.zeros(12))
    same("lfilter (python loop)", dsp.lfilter_df2t(bdec, adec, x[:700], np.zeros(12), force_python=True)[0], y_ref)
    y_c, z_c = dsp.lfilter_df2t(bdec, adec, x[:700], np.zeros(12))
    same("lfilter (C)", y_c, y_ref)
    same("lfilter state (C)", z_c, z_ref)
    iir = {"x_dec": x.astype(np.float32)}
    zr = decimate_multiple_filtic(2, bdec, adec)
    zo = dsp.decimate_multiple_filtic(2, bdec, adec)
    for c in range(4):
        chunk = x[c * 512:(c + 1) * 512]
        yr, zr = decimate_multiple(2, bdec, adec, chunk, zr)
        yo, zo = dsp.decimate_multiple(2, bdec, adec, chunk, zo)
        same(f"decimate_multiple chunk {c}", yo, yr)
        iir[f"dec2_{c}"] = yr
    for bpo in (1, 3, 6, 12, 24):
        boct = [np.array(f) for f in generated_filters.PARAMS[str(bpo)][0]]
        aoct = [np.array(f) for f in generated_filters.PARAMS[str(bpo)][1]]
        same(f"boct table {bpo}", tabs[f"boct_{bpo}"], np.array(boct))
        same(f"aoct table {bpo}", tabs[f"aoct_{bpo}"], np.array(aoct))
        xin = as_f32_f64(noise(42 + bpo, 2 * 1024))
        zr = octave_filter_bank_decimation_filtic(bdec, adec, boct, aoct)
        zo = dsp.iir_bank_filtic(bdec, adec, boct, aoct)
        iir[f"bank{bpo}_x"] = xin.astype(np.float32)
        for blk in range(2):
            yr, dr, zr = octave_filter_bank_decimation(bdec, adec, boct, aoct, xin[blk * 1024:(blk + 1) * 1024], zr)
            yo, do, zo = dsp.iir_bank(bdec, adec, boct, aoct, xin[blk * 1024:(blk + 1) * 1024], zo)
            assert dr == do
            for k in range(len(yr)):
                same(f"iir bank bpo={bpo} blk={blk} band={k}", yo[k], yr[k]) if k in (0, len(yr) - 1) else None
                assert np.array_equal(yo[k], yr[k])
            iir[f"bank{bpo}_energy_{blk}"] = np.array([np.sum(v ** 2) for v in yr])
            if bpo == 3:
                for k in range(len(yr)):
                    iir[f"bank3_y_{blk}_{k}"] = yr[k]
        iir[f"bank{bpo}_dec"] = np.array(dr)
        iir[f"bank{bpo}_zf"] = np.concatenate(zr)
    np.savez_compressed(GOLD / "iir.npz", **iir)

    # ---- O2/O3: FFT-OLA bank + band tables ---------------------------------------------------------
    print("O2/O3 Octave_Filters.filter + band tables")
    ola = {}
    for bpo in (1, 3, 6, 12, 24):
        of = Octave_Filters(bpo)
        mine = dsp.OlaBank(bpo)
        xin = as_f32_f64(noise(142 + bpo, 3 * 1024))
        ola[f"ola{bpo}_x"] = xin.astype(np.float32)
        lens = [1024, 512, 1024]           # a short block in the middle exercises the pending tails
        pos = 0
        for blk, n in enumerate(lens):
            yr, dr = of.filter(xin[pos:pos + n])
            yo, do = mine.filter(xin[pos:pos + n])
            pos += n
            assert list(dr) == list(do)
            worst = max(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300) for a, b in zip(yo, yr))
            same(f"ola bank bpo={bpo} blk={blk} (all bands)", worst, 0.0, tol=0) if False else None
            print(f"  ok  oracle vs reference: ola bank bpo={bpo} blk={blk} worst band rel err {worst:.3e}")
            assert worst < 1e-12, worst
            ola[f"ola{bpo}_energy_{blk}"] = np.array([np.sum(v ** 2) for v in yr])
            if bpo == 3:
                for k in range(len(yr)):
                    ola[f"ola3_y_{blk}_{k}"] = yr[k]
        ola[f"ola{bpo}_dec"] = np.array(dr)
        same(f"get_decs {bpo}", dsp.get_decs(bpo), of.get_decs())
        fi, flo, fhi = dsp.octave_frequencies(of.nbands, bpo)
        same(f"fi {bpo}", fi, of.fi)
        same(f"flow {bpo}", flo, of.flow)
        same(f"fhigh {bpo}", fhi, of.fhigh)
        for nm, o, r in zip("ABC", dsp.band_weighting(of.fi), (of.A, of.B, of.C)):
            same(f"band weighting {nm} {bpo}", o, r)
        ola[f"bands{bpo}_fi"], ola[f"bands{bpo}_flow"], ola[f"bands{bpo}_fhigh"] = of.fi, of.flow, of.fhigh
        ola[f"bands{bpo}_A"], ola[f"bands{bpo}_B"], ola[f"bands{bpo}_C"] = of.A, of.B, of.C
        ola[f"bands{bpo}_nominal"] = np.array(of.f_nominal)
    np.savez_compressed(GOLD / "ola.npz", **ola)

    # ---- G1: GCC-PHAT ------------------------------------------------------------------------------
    print("G1 GCC-PHAT")
    gcc = {}
    for L in (2400, 24000):
        rng = np.random.default_rng(5)
        d0 = as_f32_f64(0.25 * rng.standard_normal(L))
        d1 = as_f32_f64(np.roll(d0, 37) + 0.1 * 0.25 * rng.standard_normal(L))
        r0, r1 = d0.copy(), d1.copy()
        ref = generalized_cross_correlation(r0, r1)      # mutates r0, r1 (mean removal in place)
        mine, m0, m1 = dsp.gcc_phat(d0, d1)
        same(f"gcc L={L}", mine, ref)
        same(f"gcc in-place demean L={L}", m0, r0)
        assert int(np.argmax(np.abs(ref))) == 37
        gcc[f"L{L}_d0"], gcc[f"L{L}_d1"] = d0.astype(np.float32), d1.astype(np.float32)
        if L == 2400:
            gcc[f"L{L}_xcorr"] = ref
        else:
            gcc[f"L{L}_xcorr_head"] = ref[:128]
            gcc[f"L{L}_xcorr_norms"] = np.array([np.max(np.abs(ref)), np.sqrt(np.sum(ref ** 2)), np.std(ref)])
        gcc[f"L{L}_argmax"] = int(np.argmax(np.abs(ref)))
    np.savez_compressed(GOLD / "gcc.npz", **gcc)

    # ---- R1: ring buffer -----------------------------------------------------------------------------
    print("R1 ring buffer")
    rb, mine = RingBuffer(), dsp.MirrorRing()
    rng = np.random.default_rng(11)
    ring = {}
    for step, n in enumerate([512, 512, 7000, 512, 9000, 300]):
        blk = rng.standard_normal((1, n))
        rb.push(blk, 0.0)
        mine.push(blk)
        ln = min(rb.offset, 4096)
        same(f"ring step {step}", mine.data_indexed(mine.offset - 100, ln - 100), rb.data_indexed(rb.offset - 100, ln - 100))
        ring[f"blk{step}"] = blk
        ring[f"win{step}"] = rb.data_indexed(rb.offset - 100, ln - 100).copy()
    np.savez_compressed(GOLD / "ring.npz", **ring)

    # ---- digest of the reference's design artefacts ------------------------------------------------------
    ref_fft = np.load(Path(refshim.REFERENCE_ROOT) / "friture" / "data" / "generated_fft.npz")
    for bpo in (1, 3, 6, 12, 24):
        same(f"boct_fir table {bpo}", tabs[f"boct_fir_{bpo}"], ref_fft[f"{bpo}_boct_fir"])
        H = np.fft.rfft(tabs[f"boct_fir_{bpo}"], int(tabs["fft_sizes"][0]), axis=1)
        same(f"H_oct stage 0 {bpo}", H, ref_fft[f"{bpo}_fft_H_oct"][0][:, :H.shape[1]], tol=1e-13)
    same("bdec_fir table", tabs["bdec_fir"], ref_fft["bdec_fir"])
    digest = {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for k, v in sorted(tabs.items())}
    (GOLD / "filter_tables.sha256").write_text("".join(f"{v}  {k}\n" for k, v in digest.items()))
    print("golden fixtures written to", GOLD)
    for f in sorted(GOLD.iterdir()):
        print(f"  {f.name}: {f.stat().st_size} bytes")


if __name__ == "__main__":
    sys.exit(main())
