#!/usr/bin/env python
"""Per-push latency of the interactive chains: 1000 pushes of 512 samples (one audio chunk, 10.7 ms of signal).

    python tools/stream_latency.py > gpurun_out/stream_latency.json

For every chain: p50 / p99 / mean of the wall time of handle_new_data as the widget would call it — host buffer in, result
on the host — for the device-resident object, for the block-by-block drop-in classes (each block its own host-staged call)
and for the numpy oracle (the reference's arithmetic on this box's host, 1 core)."""
import json
import sys
import time
from fractions import Fraction
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def stats(ts):
    ts = np.sort(np.asarray(ts)) * 1e6
    return {"p50_us": float(ts[len(ts) // 2]), "p99_us": float(ts[int(len(ts) * 0.99)]), "mean_us": float(ts.mean()), "pushes": len(ts)}


def time_pushes(fn, chunks, idle=None):
    """Per-call wall time.  `idle`: called between the timed calls (outside the timed region) — for an object whose
    pushes are asynchronous, waiting for the device there gives what a caller fed at the audio rate (one chunk per
    10.7 ms) sees; back to back, the same object is bounded by the device time per chunk."""
    out = []
    for c in chunks:
        if idle is not None:
            idle()
        t0 = time.perf_counter()
        fn(c)
        out.append(time.perf_counter() - t0)
    return out


def main():
    from friture_amd import _lib
    from friture_amd.spectrogram import Spectrogram, SpectrogramStream
    from oracle import dsp
    _lib.init(0)
    rng = np.random.default_rng(0)
    n_push, chunk = 1000, 512
    x = 0.25 * rng.standard_normal((n_push + 50) * chunk)
    chunks = [x[None, i * chunk:(i + 1) * chunk] for i in range(n_push + 50)]
    res = {}
    for n_fft in (1024, 4096):
        kw = dict(fft_size=n_fft, overlap=Fraction(3, 4), weighting=1, screen_width=800, screen_height=400, timerange_s=10.0)
        dev, host = SpectrogramStream(**kw), Spectrogram(**kw)
        for name, obj in (("device_resident", dev), ("block_by_block", host)):
            time_pushes(obj.handle_new_data, chunks[:50])                 # warm-up: allocations, first launches
            res[f"spectrogram_N{n_fft}_{name}"] = stats(time_pushes(obj.handle_new_data, chunks[50:]))
        # the import swap proper: the widget's own body (friture/spectrogram.py:131-177) on the swapped classes —
        # host RingBuffer, audioproc.analyzelive PER FRAME, numpy log / normalise, Transform_Pipeline.push
        from friture_amd.audioproc import audioproc as Proc
        from friture_amd.ringbuffer import RingBuffer
        from friture_amd.signal.color_tranform import Color_Transform
        from friture_amd.signal.frequency_resampler import Frequency_Resampler
        from friture_amd.signal.online_linear_2D_resampler import Online_Linear_2D_resampler
        from friture_amd.signal.transform_pipeline import Transform_Pipeline
        from friture_amd.plotting import frequency_scales as fsc
        proc = Proc()
        proc.set_fftsize(n_fft)
        fr, trs = Frequency_Resampler(fsc.Mel, 20., 20000., 400), Online_Linear_2D_resampler()
        fr.setfreq(proc.get_freq_scale())
        pipe = Transform_Pipeline([fr, trs, Color_Transform()])
        wA = proc.get_freq_weighting()[0]
        rb, stw = RingBuffer(), {"old": 0}
        sfft_w = Fraction(48000, n_fft) / (Fraction(1) - Fraction(3, 4)) / 1000
        needed_w = n_fft * (1. - 0.75)

        def widget_push(c):
            rb.push(c, 0.)
            available = rb.offset - stw["old"]
            realizable = int(np.floor(available / needed_w))
            if realizable <= 0:
                return None
            spn = np.zeros((n_fft // 2 + 1, realizable), dtype=np.float64)
            for i in range(realizable):
                floatdata = rb.data_indexed(stw["old"], n_fft)
                spn[:, i] = proc.analyzelive(floatdata[0, :])
                stw["old"] += int(needed_w)
            w = np.tile(wA, (realizable, 1)).transpose()
            norm = (10. * np.log10(spn + 1e-30) + w - (-140.)) / 140.
            trs.set_height(400)
            trs.set_ratio(sfft_w, Fraction(800, 10000))
            fr.setnsamples(400)
            return pipe.push(norm)

        time_pushes(widget_push, chunks[:50])
        res[f"spectrogram_N{n_fft}_import_swap"] = stats(time_pushes(widget_push, chunks[50:]))
        # the oracle's chain (numpy float64, the reference's arithmetic)
        lut = dsp.colour_lut(dsp.cmrmap())
        w = dsp.weighting_curves(dsp.frequency_axis(n_fft))[0]
        tg = dsp.frequency_targets("mel", 20.0, 20000.0, 400)
        sfft = Fraction(48000, n_fft) / (Fraction(1) - Fraction(3, 4)) / 1000
        tr = dsp.TimeResampler(sfft, Fraction(800, 10000), 400)
        ring, state = dsp.MirrorRing(), {"old": 0}
        hop, win, fax = n_fft // 4, dsp.hann_symmetric(n_fft), dsp.frequency_axis(n_fft)

        def oracle_push(c):
            ring.push(c)
            realizable = int(np.floor((ring.offset - state["old"]) / float(hop)))
            if realizable <= 0:
                return None
            cols = []
            for _ in range(realizable):
                cols.append(dsp.psd_frame(ring.data_indexed(state["old"], n_fft)[0], win))
                state["old"] += hop
            norm = dsp.normalise(dsp.log_spectrum(np.stack(cols, axis=1)) + w[:, None], -140.0, 0.0)
            return dsp.colour_pixels(lut, tr.push(dsp.frequency_resample(tg, fax, norm)))[::-1, :]

        time_pushes(oracle_push, chunks[:50])
        res[f"spectrogram_N{n_fft}_numpy_oracle"] = stats(time_pushes(oracle_push, chunks[50:250]))
    # ---- spectrum widget: N = 8192 (its default), 75 % overlap ------------------------------------------------------------
    from friture_amd.spectrum import SpectrumAnalyzer, SpectrumAnalyzerStream
    for name, obj in (("device_resident", SpectrumAnalyzerStream()), ("block_by_block", SpectrumAnalyzer())):
        time_pushes(obj.handle_new_data, chunks[:50])
        res[f"spectrum_N8192_{name}"] = stats(time_pushes(obj.handle_new_data, chunks[50:]))
    sa = SpectrumAnalyzer()
    ring3, st3 = dsp.MirrorRing(), {"old": 0, "prev": np.zeros(4097)}
    w8 = dsp.weighting_curves(dsp.frequency_axis(8192))[0]
    win8, fax8 = dsp.hann_symmetric(8192), dsp.frequency_axis(8192)

    def oracle_spectrum(c):
        ring3.push(c)
        realizable = int(np.floor((ring3.offset - st3["old"]) / 2048.0))
        if realizable <= 0:
            return None
        cols = []
        for _ in range(realizable):
            cols.append(dsp.psd_frame(ring3.data_indexed(st3["old"], 8192)[0], win8))
            st3["old"] += 2048
        r = dsp.spectrum_readout(np.stack(cols, axis=1), sa.kernel, sa.alpha, st3["prev"], w8, fax8)
        st3["prev"] = r["smoothed"]
        return r

    time_pushes(oracle_spectrum, chunks[:50])
    res["spectrum_N8192_numpy_oracle"] = stats(time_pushes(oracle_spectrum, chunks[50:450]))
    # ---- octave spectrum: 1/3 and 1/24 octave -------------------------------------------------------------------
    from friture_amd.octavespectrum import OctaveSpectrum, OctaveSpectrumStream
    xf = [c.astype(np.float32).astype(np.float64) for c in chunks]
    for bpo in (3, 24):
        dev, host = OctaveSpectrumStream(bpo, 1, 1.0), OctaveSpectrum(bpo, 1, 1.0)
        for name, obj in (("device_resident", dev), ("block_by_block", host)):
            time_pushes(obj.handle_new_data, xf[:50])
            res[f"octave_bpo{bpo}_{name}"] = stats(time_pushes(obj.handle_new_data, xf[50:]))
        bank = dsp.OlaBank(bpo)
        alphas, kernels = dsp.band_smoothing_setup(bpo, 1.0)
        fi, _, _ = dsp.octave_frequencies(9 * bpo, bpo)
        A = dsp.band_weighting(fi)[0]
        st = {"prev": [0.0] * (9 * bpo)}

        def oracle_oct(c):
            y, _ = bank.filter(c[0])
            st["prev"] = dsp.band_energies(y, kernels, alphas, st["prev"])
            return dsp.band_db(np.array(st["prev"]), A)

        time_pushes(oracle_oct, xf[:20])
        res[f"octave_bpo{bpo}_numpy_oracle"] = stats(time_pushes(oracle_oct, xf[50:250]))
    # ---- delay estimator: 2 channels, default 1 s range (24000-sample windows every 12000 decimated samples) ------------
    from friture_amd.delay_estimator import DelayEstimator, DelayEstimatorStream
    from friture_amd import filter_design
    t = filter_design.load_tables()
    x2 = [np.stack([c[0], np.roll(c[0], 40)]) for c in xf]
    for name, obj in (("device_resident", DelayEstimatorStream(1.0)), ("block_by_block", DelayEstimator(1.0))):
        time_pushes(obj.handle_new_data, x2[:50])
        res[f"delay_{name}"] = stats(time_pushes(obj.handle_new_data, x2[50:]))
    import torch
    obj = DelayEstimatorStream(1.0)
    time_pushes(obj.handle_new_data, x2[:50])
    res["delay_device_resident_paced"] = stats(time_pushes(obj.handle_new_data, x2[50:], idle=torch.cuda.synchronize))
    bdec, adec = np.array(t["bdec"]), np.array(t["adec"])
    z = [dsp.decimate_multiple_filtic(2, bdec, adec), dsp.decimate_multiple_filtic(2, bdec, adec)]
    rings, st2 = [dsp.MirrorRing(), dsp.MirrorRing()], {"old_index": 0, "old": None}

    def oracle_delay(c):
        for ch in range(2):
            d, z[ch] = dsp.decimate_multiple(2, bdec, adec, c[ch], z[ch])
            rings[ch].push(d[None, :])
        avail = rings[0].offset - st2["old_index"]
        for _ in range(int(avail / 12000)):
            st2["old_index"] += 12000
            d0 = rings[0].data_indexed(st2["old_index"], 24000).reshape(-1)
            d1 = rings[1].data_indexed(st2["old_index"], 24000).reshape(-1)
            xc, _, _ = dsp.gcc_phat(d0, d1)
            ro = dsp.delay_readout(xc, st2["old"], 12000.0, 1.0)
            st2["old"] = ro["smoothed"]

    time_pushes(oracle_delay, x2[:50])
    res["delay_numpy_oracle"] = stats(time_pushes(oracle_delay, x2[50:450]))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
