"""Stress: the LDS-DMA instances (hand-counted waits) on long signals, repeated: every repetition must give the same bits (a
race on the staging buffers would not), the N = 1024 ring instance the bits of the register-window instance, and the
large-frame instances the register-path instance's spectra to rounding (their window multiply fuses differently)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import torch
from friture_amd import _lib
from friture_amd.stft import StftEngine
_lib.init(0)
gen = torch.Generator(device="cuda").manual_seed(3)
bad = 0
for n_fft, hop, log2t in ((1024, 512, 24), (4096, 2048, 24), (4096, 1024, 23), (8192, 4096, 24), (16384, 8192, 24), (16384, 4096, 23)):
    T = 1 << log2t
    buf = 0.25 * torch.randn(T + 4, generator=gen, device="cuda", dtype=torch.float32)
    e = StftEngine(n_fft, hop, 1, 32)
    shifted = buf[1:T + 1].reshape(1, T)                 # off the 16-byte grid: register-path instance
    aligned = shifted.clone().contiguous()               # same samples on the grid: LDS-DMA instance
    assert shifted.data_ptr() % 16 != 0 and aligned.data_ptr() % 16 == 0
    ref = e.psd(shifted)
    first = e.psd(aligned).clone()
    scale = ref.amax(dim=2, keepdim=True)
    rel = float(((first - ref).abs() / scale).max())
    if n_fft == 1024 and not torch.equal(first, ref):
        bad += 1
        print("RING != WINDOW", int((first != ref).sum()))
    if rel > 2e-6:
        bad += 1
        print("TOO FAR FROM THE REGISTER-PATH INSTANCE", n_fft, hop, rel)
    for rep in range(12):
        got = e.psd(aligned)
        if not torch.equal(got, first):
            bad += 1
            print("NOT REPRODUCIBLE", n_fft, hop, rep, int((got != first).sum()))
    print(n_fft, hop, "frames", ref.shape[1], "max rel diff to the register-path instance %.2e" % rel)
print("stress done, mismatching runs:", bad)
