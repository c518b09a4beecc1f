"""Band energies of the time-parallel IIR bank: contracted multiply-adds (default for energy-only calls) against the
reference's separately rounded operations (FRT_IIR_EXACT_OPS=1) and against the bit-exact sequential mode."""
import os, subprocess, sys, json
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
code = '''
import sys; sys.path.insert(0, %r)
import numpy as np, torch
from friture_amd import _lib, filter_design
from friture_amd.filter import IirBank
_lib.init(0)
t = filter_design.load_tables()
out = {}
for bpo in (3, 24):
    ch, n = 2, 1 << 19
    rng = np.random.default_rng(5)
    x = torch.from_numpy((0.25 * rng.standard_normal((ch, n))).astype(np.float32)).cuda()
    decs = [2 ** j for j in range(9)[::-1] for _ in range(bpo)]
    alphas = np.array([1.0 - (1.0 - 0.65) ** (1.0 / (1.0 * 48000 / d + 1)) for d in decs])
    for chunk in (2048, 0):
        b = IirBank(t["bdec"], t["adec"], list(t[f"boct_{bpo}"]), list(t[f"aoct_{bpo}"]), ch)
        b.set_chunk(chunk)
        e = b.energies(x, 1024, alphas)
        out[f"{bpo}_{chunk}"] = e.double().cpu().numpy().tolist()
import json; print(json.dumps(out))
''' % str(ROOT)
res = {}
for tag, env in (("fused", {}), ("exact", {"FRT_IIR_EXACT_OPS": "1"})):
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env={**os.environ, **env})
    res[tag] = json.loads(r.stdout.strip().splitlines()[-1])
import numpy as np
for bpo in (3, 24):
    f, e, s = (np.array(res["fused"][f"{bpo}_2048"]), np.array(res["exact"][f"{bpo}_2048"]), np.array(res["exact"][f"{bpo}_0"]))
    print(f"bpo {bpo}: fused vs exact-ops time-parallel {np.max(np.abs(f / e - 1)):.2e}; fused vs sequential (bit-exact) {np.max(np.abs(f / s - 1)):.2e}; "
          f"exact-ops time-parallel vs sequential {np.max(np.abs(e / s - 1)):.2e}   (float32 outputs; bar 1e-5)")
