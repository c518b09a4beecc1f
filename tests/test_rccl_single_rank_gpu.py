"""RCCL itself, on the one GPU a test box has: a process group of ONE rank on the nccl backend (= RCCL on ROCm) with
FRT_DIST_FORCE=1, so that every helper of friture_amd/distributed.py goes through the collective library with device tensors
— communicator creation with `device_id`, broadcast of the constant tables (float64 and the uint32 LUT as an int32 view),
all_gather of summaries / scalars / rank ids, all_reduce(MAX), barrier(device_ids), and the asynchronous
all_gather_into_tensor of SlabGather behind a kernel of the product on the current stream.  What one GPU cannot show is the
exchange between ranks: tests/test_distributed_cpu.py covers the two-rank logic over gloo, tests/test_sharding_gpu.py that a
rank's results do not depend on the other channels of the launch."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]

SCRIPT = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["FRT_ROOT"])
from friture_amd import distributed, _lib
from friture_amd.stft import StftEngine

rank, local_rank, world = distributed.init_process_group()
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1, (dist.is_initialized(), world)
dev = torch.device("cuda", local_rank)
_lib.init(local_rank)
assert distributed.gather_ranks(dev) == [0]
consts = {"weight": np.linspace(-3.0, 1.0, 513), "lut": (np.arange(256, dtype=np.uint32) * 0x01010101).astype(np.uint32)}
got = distributed.broadcast_tables(consts, src=0, device=dev)
assert got["weight"].dtype == np.float64 and np.array_equal(got["weight"], consts["weight"])
assert got["lut"].dtype == np.uint32 and np.array_equal(got["lut"], consts["lut"])
# a kernel of the product on the current stream, then the slab gather ordered behind it
n_fft, hop, ch, T = 1024, 512, 2, 1 << 16
eng = StftEngine(n_fft, hop, ch, 32)
x = torch.randn((ch, T), device=dev, dtype=torch.float32) * 0.25
F = eng.frames_for(T)
out = torch.empty((ch, F, n_fft // 2 + 1), dtype=torch.float32, device=dev)
eng.run(0, x, out)
sg = distributed.SlabGather(out, n_slots=2)
sg.start(out, 0)
eng.run(0, x, out.clone())            # the next batch overlaps the gather
g = sg.wait(0)
torch.cuda.synchronize()
assert tuple(g.shape) == (1, ch, F, n_fft // 2 + 1) and torch.equal(g[0], out)
summ = out.sum(dim=(1, 2), dtype=torch.float64)[:, None]
allsumm = distributed.gather_channel_summaries(summ, ch)
assert allsumm.shape == (ch, 1) and torch.equal(allsumm, summ)
assert distributed.gather_scalars(1.25, dev) == [1.25]
assert distributed.max_over_ranks(0.5, dev) == 0.5
distributed.barrier(dev)
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK")
"""


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_every_distributed_helper_on_rccl_with_one_rank():
    env = dict(os.environ, FRT_ROOT=str(ROOT), FRT_DIST_FORCE="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29671", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.gpu
@pytest.mark.timeout(400)
def test_bench_under_torchrun_on_rccl_with_one_rank():
    """bench.py launched the way the contract launches it for N > 1 (python -m torch.distributed.run ... bench.py --gpus N), with
    N = 1 and FRT_DIST_FORCE=1: the table broadcast, rank gather, barriers, max-over-ranks and the optional slab gather of the
    bench all run on RCCL with the product's kernels between them; the line must carry ranks_seen = [0] and a verified gather."""
    import json
    env = dict(os.environ, FRT_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29673", str(ROOT / "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--no-legs",
           "--cpu-budget", "0", "--log2-samples", "22", "--gather-slabs"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=380, cwd=str(ROOT))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == [0] and line["parity"]["gate"]["pass"]
    assert line["slab_gather"]["enabled"] and line["slab_gather"]["slabs_verified"]
