"""N > 1 plumbing on CPU: world_size 2 over gloo (the GPU path is the same code over RCCL)."""
import os
import socket

import numpy as np
import pytest

from friture_amd import distributed


def test_shard_channels_partition():
    for n, w in [(256, 8), (64, 8), (8, 8), (3, 2), (1, 4), (10, 4)]:
        owned = [c for r in range(w) for c in distributed.shard_channels(n, r, w)]
        assert owned == list(range(n))
        sizes = [len(distributed.shard_channels(n, r, w)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    r, lr, w = distributed.init_process_group(backend="gloo")
    assert (r, w) == (rank, world)
    # plan-time broadcast: only rank 0 holds the real tables
    tabs = {"weight": np.zeros(513), "lut": np.zeros(256, np.uint32)}
    if rank == 0:
        tabs = {"weight": np.linspace(-30, 2, 513), "lut": (np.arange(256, dtype=np.uint32) * 0x010101) | 0xFF000000}
    tabs = distributed.broadcast_tables(tabs)
    assert np.array_equal(tabs["weight"], np.linspace(-30, 2, 513))
    assert tabs["lut"].dtype == np.uint32 and tabs["lut"][255] == 0xFFFFFFFF
    # channel-sharded "batch": 5 channels over 2 ranks, summary row = channel id and its square
    n_channels = 5
    mine = distributed.shard_channels(n_channels, rank, world)
    local = torch.tensor([[float(c), float(c * c)] for c in mine], dtype=torch.float64).reshape(len(mine), 2)
    full = distributed.gather_channel_summaries(local, n_channels)
    assert full.shape == (5, 2) and torch.equal(full[:, 0], torch.arange(5, dtype=torch.float64))
    assert torch.equal(full[:, 1], torch.arange(5, dtype=torch.float64) ** 2)
    assert distributed.max_over_ranks(1.0 + rank) == float(world)
    distributed.barrier()
    dist.destroy_process_group()
    q.put(rank)


def test_two_rank_gloo_roundtrip():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]
