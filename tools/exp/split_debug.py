"""round 5: where do split rows differ from packed rows at full size?  (tests/test_stft_gpu.py::test_split_rows_full_size_headline_image)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from friture_amd import _lib, palette, tables
from friture_amd.stft import StftEngine
_lib.init(0)
n_fft, hop = 1024, 512
weight = tables.weighting_db(tables.rfft_frequencies(n_fft), 1e-50)[0]
for log2t in (20, 23, 26):
    T = 1 << log2t
    x = torch.from_numpy((0.25 * np.random.default_rng(42).standard_normal(T, dtype=np.float32))[None]).cuda()
    e = StftEngine(n_fft, hop, 1, 32)
    e.set_epilogue(weight, -140.0, 0.0, palette.cmr_lut())
    for kind in (3, 0):
        p1 = e.run(kind, x).view(torch.int32)
        p2 = e.run(kind, x).view(torch.int32)
        rows, nyq = e.run_split(kind, x)
        rows2, nyq2 = e.run_split(kind, x)
        torch.cuda.synchronize()
        rows, nyq, rows2 = rows.view(torch.int32), nyq.view(torch.int32), rows2.view(torch.int32)
        bad = rows != p1[..., :512]
        print(f"T=2^{log2t} kind={kind}: packed twice equal {torch.equal(p1, p2)}; split twice equal {torch.equal(rows, rows2)}; "
              f"rows differing {int(bad.sum())} of {bad.numel()}; nyq differing {int((nyq != p1[..., 512]).sum())}")
        if bad.any():
            idx = torch.nonzero(bad[0])
            f, k = idx[:, 0].cpu().numpy(), idx[:, 1].cpu().numpy()
            print("   frames:", f[:12], "... bins:", k[:12], " frames mod 16:", np.bincount(f % 16, minlength=16), " distinct frames", len(np.unique(f)))
            print("   bins histogram (64-wide):", np.bincount(k // 64, minlength=8))
            a, b = rows[0, f[0], k[0]].item(), p1[0, f[0], k[0]].item()
            print(f"   first: split {a:#x} packed {b:#x}")
