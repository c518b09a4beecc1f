"""Channel sharding across the GPUs of one node (one process per GPU, torch.distributed / RCCL).

Audio channels are independent everywhere on the hot path (the reference processes channel 0 and
optionally channel 1 separately: friture/spectrum.py:149-153, friture/delay_estimator.py:97-98),
so the channel axis is block-partitioned across ranks and the data path needs no collective.
Two collectives exist around it, both tiny:

  * plan time: rank 0 broadcasts the constant tables (weighting curve, colour LUT, filter
    coefficients) so that every rank computes with bit-identical constants;
  * after a batch: per-channel summaries (band energies, spectrogram digests) are all-gathered so
    that rank 0 can present all channels.  Full PSD / pixel slabs stay resident on the GPU that
    produced them (an all-gather of those is per-link bound on xGMI and nobody consumes it whole).

The same code runs on CPU tensors over gloo (tests/test_distributed_cpu.py, world_size 2).
"""
from __future__ import annotations

import os

import numpy as np


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when not launched by it."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_channels(n_channels: int, rank: int, world: int) -> range:
    """Block partition: rank r owns channels [r*C/W, (r+1)*C/W) (remainder spread over low ranks)."""
    base, extra = divmod(n_channels, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def init_process_group(backend: str | None = None, device=None):
    """Join the job's process group (RCCL on GPU, gloo on CPU).  No-op for single-process runs."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if world == 1 or dist.is_initialized():
        return rank, local_rank, world
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kwargs = {}
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        kwargs["device_id"] = torch.device("cuda", local_rank)
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def broadcast_tables(tables: dict, src: int = 0, device=None) -> dict:
    """Broadcast a dict of numpy arrays from `src` (shapes / dtypes must be known on all ranks)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return tables
    out = {}
    for key in sorted(tables):
        arr = np.ascontiguousarray(tables[key])
        view = arr.view(np.int32) if arr.dtype == np.uint32 else arr      # same bits, torch-friendly dtype
        t = torch.from_numpy(view.copy())
        if device is not None:
            t = t.to(device)
        dist.broadcast(t, src=src)
        got = t.cpu().numpy()
        out[key] = got.view(np.uint32) if arr.dtype == np.uint32 else got
    return out


def gather_channel_summaries(local, n_channels: int):
    """All-gather per-channel summary rows.  `local`: tensor [local_channels, K] of this rank's
    shard (block partition of `n_channels`).  Returns [n_channels, K] on every rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    width = local.shape[1]
    most = -(-n_channels // world)
    padded = torch.zeros((most, width), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    rows = [parts[r][: len(shard_channels(n_channels, r, world))] for r in range(world)]
    return torch.cat(rows, dim=0)


def gather_ranks(device=None) -> list:
    """The rank ids an all-gather over the job's process group returns ([0] for a single process): evidence that the
    collective library (RCCL on GPUs) saw every rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [0]
    mine = torch.tensor([dist.get_rank()], dtype=torch.int64, device=device)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [int(p.item()) for p in parts]


def max_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(device=None):
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        if device is not None and getattr(device, "type", "cpu") == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()
