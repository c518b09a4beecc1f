"""GCC-PHAT batch throughput (kernel K5): windows/s against the batch size, the launch shapes (one workgroup per pair: the resident kernel and the one with the scratch slab /
a pair's sub-transforms as workgroups of their own), device-resident float64 windows of L samples.

    python tools/bench_gcc.py [--length 24000] [--pairs 1 4 16 32 64 100 256 1024]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch

    from friture_amd import _lib
    from friture_amd.signal.correlation import GccPhat
    ap = argparse.ArgumentParser()
    ap.add_argument("--length", type=int, default=24000)
    ap.add_argument("--pairs", type=int, nargs="*", default=[1, 4, 16, 32, 64, 100, 256, 1024])
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    L = args.length
    dev = torch.device("cuda", 0)
    res = {}
    for pairs in args.pairs:
        rng = np.random.default_rng(pairs)
        d0 = 0.25 * rng.standard_normal((pairs, L))
        d1 = np.roll(d0, 37, axis=1) + 0.025 * rng.standard_normal((pairs, L))
        a0, a1 = torch.from_numpy(d0).to(dev), torch.from_numpy(d1).to(dev)
        row = {}
        for shape, opt, resident in (("default", -1, -1), ("one_workgroup", 1, -1), ("one_workgroup_slab", 1, 0), ("split", 0, -1)):
            _lib.set_option("gcc_one_workgroup", opt)
            _lib.set_option("gcc_resident", resident)
            g = GccPhat(L, pairs)
            for _ in range(3):
                _, am = g.correlate(a0, a1)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                g.correlate(a0, a1)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            row[shape] = {"ms": ms, "windows_per_s": pairs / ms * 1e3, "delay_found": bool(int(am[0]) == 37)}
        _lib.set_option("gcc_one_workgroup", -1)
        _lib.set_option("gcc_resident", -1)
        res[str(pairs)] = row
        print(pairs, {k: (round(v["ms"], 4), round(v["windows_per_s"])) for k, v in row.items()}, flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
