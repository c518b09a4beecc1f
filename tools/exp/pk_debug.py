"""Parity probe of the packed N = 16384 instance (stft_pk.h) against the oracle, with the error's structure when it fails:
per frame, and for the worst frame per bin class (k mod 16 = LDS region, k // 512 = unpack slot q, k mod 512 = thread)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
from friture_amd import _lib
from friture_amd.stft import StftEngine
from oracle import dsp
_lib.init(0)
rng = np.random.default_rng(7)
bad = 0
for n_fft, hop, frames, C, run in ((16384, 8192, 7, 2, 0), (16384, 4096, 11, 2, 0), (16384, 8192, 5, 1, 3), (16384, 4096, 9, 3, 5)):
    T = n_fft + hop * (frames - 1) + 8
    x = (0.25 * rng.standard_normal((C, T))).astype(np.float32)
    e = StftEngine(n_fft, hop, C, 32)
    e.set_run_length(run)
    got = e.psd(x)
    for c in range(C):
        ref = dsp.stft_psd(x[c].astype(np.float64), n_fft, hop)
        err = np.max(np.abs(got[c] - ref), axis=1) / np.max(ref, axis=1)
        print(f"N {n_fft} hop {hop} run {run} ch {c}: per-frame rel err", " ".join(f"{v:.1e}" for v in err))
        if np.max(err) > 1e-5:
            bad += 1
            f = int(np.argmax(err > 1e-5))
            d = np.abs(got[c][f] - ref[f]) / np.max(ref[f])
            k = np.arange(len(d))
            wrong = d > 1e-5
            print(f"  frame {f}: {wrong.sum()} of {len(d)} bins wrong; first wrong bins {k[wrong][:12]}")
            print("  wrong by k % 16 :", np.bincount(k[wrong] % 16, minlength=16))
            print("  wrong by k // 512:", np.bincount(k[wrong] // 512, minlength=17))
            tt = k[wrong] % 512
            print("  wrong by thread // 64:", np.bincount(tt // 64, minlength=8), " thread % 64 range", tt.min() % 64, tt.max() % 64)
            print("  got/ref at first wrong:", got[c][f][k[wrong][:4]], ref[f][k[wrong][:4]])
print("pk_debug: bad (channel, config) pairs:", bad)
