"""stft_pk16r_kernel against stft_pk16_kernel on BASELINE configs[3]'s shard (32 ch x 2^20, N = 16384), bin by bin, on a
-DFRT_EXPERIMENTS variant library (FRT_LIB_VARIANT=px): where do they differ — by frame in the run, bin class, channel."""
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from friture_amd import _lib
_lib.LIB_PATH = ROOT / "tools" / "variants" / os.environ.get("FRT_LIB_VARIANT", "px") / "libfriture_hip.so"
from friture_amd.stft import StftEngine
_lib.init(0)
hop = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
C, T, N = 32, 1 << 20, 16384
gen = torch.Generator(device="cuda").manual_seed(43)
x = 0.25 * torch.randn((C, T), generator=gen, device="cuda", dtype=torch.float32)
for kind in ("psd", "db"):
    e = StftEngine(N, hop, C, 32)
    if kind == "db":
        from friture_amd import tables
        A = tables.weighting_db(tables.rfft_frequencies(N), 1e-50)[0]
        e.set_epilogue(A, -140.0, 0.0, np.arange(256, dtype=np.uint32))
    f = getattr(e, kind)
    os.environ.pop("FRT_STFT_PK16R", None)
    ref = f(x).clone()
    os.environ["FRT_STFT_PK16R"] = "1"
    outs = [f(x).clone() for _ in range(3)]
    torch.cuda.synchronize()
    print(kind, "repeatable:", all(torch.equal(outs[0], o) for o in outs[1:]))
    got = outs[0]
    scale = ref.abs().amax(dim=2, keepdim=True) if kind == "psd" else torch.ones_like(ref[..., :1])
    d = ((got - ref).abs() / scale)
    tol = 2e-6 if kind == "psd" else 1e-3
    bad = d > tol
    print(kind, "max rel diff", float(d.max()), "bad bins", int(bad.sum()), "of", bad.numel())
    if int(bad.sum()):
        c, fr, k = torch.nonzero(bad, as_tuple=True)
        c, fr, k = c.cpu().numpy(), fr.cpu().numpy(), k.cpu().numpy()
        print("  channels:", np.bincount(c, minlength=C))
        print("  frame in run (run = %d):" % 8, np.bincount(fr % 8, minlength=8))
        print("  frames:", np.unique(fr)[:40])
        print("  bins (first 24):", np.unique(k)[:24], " count of distinct bins", len(np.unique(k)))
        print("  k mod 16:", np.bincount(k % 16, minlength=16))
        print("  k // 512:", np.bincount(k // 512, minlength=17))
        i = 0
        print("  example:", c[i], fr[i], k[i], float(got[c[i], fr[i], k[i]]), float(ref[c[i], fr[i], k[i]]))
