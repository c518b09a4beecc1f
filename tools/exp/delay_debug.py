import numpy as np, sys
sys.path.insert(0, ".")
from friture_amd.delay_estimator import DelayEstimator, DelayEstimatorStream
for rng_s, check in ((0.3, False), (1.0, True), (1.0, False)):
    a, b = DelayEstimator(rng_s), DelayEstimatorStream(rng_s)
    rng = np.random.default_rng(int(rng_s * 100))
    n = int(48000 * rng_s * 5.2)
    x0 = (0.25 * rng.standard_normal(n) + 0.01).astype(np.float32)
    x1 = np.roll(x0, 4 * 31) - 0.02 + (0.01 * rng.standard_normal(n)).astype(np.float32)
    bad = 0
    last = None
    for pos in range(0, n - 512, 512):
        chunk = np.stack([x0[pos:pos + 512], x1[pos:pos + 512]]).astype(np.float64)
        a.handle_new_data(chunk)
        b.handle_new_data(chunk)
        ne = []
        if check:
            L = 9000
            ref = np.concatenate([a.ringbuffer0.data_indexed(b.offset, L), a.ringbuffer1.data_indexed(b.offset, L)])
            got = b.window(b.offset, L)
            ne = np.argwhere(ref != got)
        cur = (a.Xcorr_extremum, b.Xcorr_extremum)
        if cur != last or len(ne):
            print(rng_s, check, pos, b.offset, a.ringbuffer0.buffer_length, cur, a.delay_ms, b.delay_ms, len(ne), ne[:2].tolist() if len(ne) else "", ne[-1:].tolist() if len(ne) else "")
            last = cur
            bad += len(ne) > 0
        if bad > 6:
            break
