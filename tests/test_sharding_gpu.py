"""Channel sharding on ONE GPU: every engine run on the channel blocks `shard_channels(C, r, W)` of W = 1, 2, 4 ranks,
in-process, and the concatenation compared with the unsharded run bit for bit.  (The 8-GPU scaling run is the driver's;
this pins what it relies on: a rank's results do not depend on which other channels share its launch.)"""
import numpy as np
import pytest

from conftest import synth
from friture_amd.distributed import shard_channels

pytestmark = pytest.mark.gpu


def shards(C, W):
    return [shard_channels(C, r, W) for r in range(W)]


@pytest.mark.parametrize("n_fft,hop,kind", [(1024, 512, "image"), (1024, 256, "psd"), (4096, 1024, "image"), (16384, 8192, "psd"), (256, 100, "db")])
def test_stft_sharded_equals_unsharded(hip, golden, n_fft, hop, kind):
    from friture_amd import tables
    from friture_amd.stft import StftEngine
    C, frames = 8, 40
    T = n_fft + hop * (frames - 1)
    x = np.stack([synth(("noise", "tone", "chirp")[c % 3], T, 300 + c) for c in range(C)])
    w = tables.weighting_db(tables.rfft_frequencies(n_fft), 1e-50)[0]
    lut = golden("image")["lut"]

    def run(block):
        e = StftEngine(n_fft, hop, block.shape[0], 32)
        e.set_epilogue(w, -140.0, 0.0, lut)
        return getattr(e, kind)(block)

    full = run(x)
    for W in (2, 4):
        parts = [run(x[list(s)]) for s in shards(C, W)]
        assert np.array_equal(np.concatenate(parts, axis=0), full), (W,)


def test_iir_bank_sharded_equals_unsharded(hip):
    from friture_amd import filter_design
    from friture_amd.filter import IirBank
    t = filter_design.load_tables()
    C, n, bpo = 8, 8192, 3
    x = np.stack([synth("noise", n, 500 + c) for c in range(C)])
    decs = [2 ** j for j in range(9)[::-1] for _ in range(bpo)]
    alphas = np.array([1.0 - (1.0 - 0.65) ** (1.0 / (0.125 * 48000 / d + 1)) for d in decs])
    for chunk in (0, 2048):
        def run(block):
            b = IirBank(t["bdec"], t["adec"], list(t[f"boct_{bpo}"]), list(t[f"aoct_{bpo}"]), block.shape[0])
            b.set_chunk(chunk)
            return b.energies(block, 1024, alphas)
        full = run(x)
        for W in (2, 4):
            assert np.array_equal(np.concatenate([run(x[list(s)]) for s in shards(C, W)], axis=0), full), (chunk, W)


def test_fir_bank_sharded_equals_unsharded(hip):
    from friture_amd.filter import FirBank
    C, n, bpo = 4, 6 * 1024, 3
    x = np.stack([synth("noise", n, 700 + c) for c in range(C)])
    decs = [2 ** j for j in range(9)[::-1] for _ in range(bpo)]
    alphas = np.array([1.0 - (1.0 - 0.65) ** (1.0 / (0.125 * 48000 / d + 1)) for d in decs])
    full = FirBank(bpo, C).energies(x, 1024, alphas)
    for W in (2, 4):
        parts = [FirBank(bpo, len(s)).energies(x[list(s)], 1024, alphas) for s in shards(C, W)]
        assert np.array_equal(np.concatenate(parts, axis=0), full), (W,)


def test_gcc_pairs_sharded_equals_unsharded(hip):
    from friture_amd.signal.correlation import GccPhat
    P, L = 8, 2400
    rng = np.random.default_rng(9)
    d0 = 0.25 * rng.standard_normal((P, L))
    d1 = np.roll(d0, 11, axis=1) + 0.02 * rng.standard_normal((P, L))
    xc, am = GccPhat(L, P).correlate(d0.copy(), d1.copy())
    for W in (2, 4):
        parts = [GccPhat(L, len(s)).correlate(d0[list(s)].copy(), d1[list(s)].copy()) for s in shards(P, W)]
        assert np.array_equal(np.concatenate([np.asarray(p[0]) for p in parts], axis=0), np.asarray(xc)), (W,)
        assert np.array_equal(np.concatenate([np.asarray(p[1]) for p in parts], axis=0), np.asarray(am)), (W,)
