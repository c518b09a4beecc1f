"""Where a 512-sample push of the block-by-block delay estimator spends its microseconds: the Python wrapper of
decimate_multiple_channels, the C call inside it (frt_decimate_multiple_state: staging, two dependent stage launches, copy back,
one wait), the rings.  p50 over 2000 calls each."""
import ctypes
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
from friture_amd import _lib, filter_design
from friture_amd.signal import decimate as D
from friture_amd.delay_estimator import DelayEstimator

lib = _lib.init(0)
t = filter_design.load_tables()
bdec, adec = np.array(t["bdec"]), np.array(t["adec"])
rng = np.random.default_rng(0)
X = 0.25 * rng.standard_normal((2, 512))
zis = [D.decimate_multiple_filtic(2, bdec, adec), D.decimate_multiple_filtic(2, bdec, adec)]


def p50(fn, n=2000):
    for _ in range(50):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return 1e6 * float(np.median(ts))


print("decimate_multiple_channels (wrapper + C call): %.1f us" % p50(lambda: D.decimate_multiple_channels(2, bdec, adec, X, zis)))
h = D._chain_handle(bdec, adec, 2)
out = np.empty((2, 128))
zi = np.zeros((2, 2, 12))
zf = np.empty((2, 2, 12))
n_out = ctypes.c_int(0)
print("frt_decimate_multiple_state alone: %.1f us" %
      p50(lambda: lib.frt_decimate_multiple_state(h, 2, X.ctypes.data, 512, zi.ctypes.data, out.ctypes.data, ctypes.byref(n_out), zf.ctypes.data)))
for n in (64, 128, 256, 1024, 2048):
    Xn = 0.25 * rng.standard_normal((2, n))
    outn = np.empty((2, n // 4))
    print("  the same with %4d samples per channel: %.1f us" % (n, p50(
        lambda: lib.frt_decimate_multiple_state(h, 2, Xn.ctypes.data, n, zi.ctypes.data, outn.ctypes.data, ctypes.byref(n_out), zf.ctypes.data), 500)))
est = DelayEstimator(1.0)
print("DelayEstimator.handle_new_data (p50, most pushes have no window): %.1f us" % p50(lambda: est.handle_new_data(X)))
