"""Channel sharding across the GPUs of one node (one process per GPU, torch.distributed / RCCL).

Audio channels are independent everywhere on the hot path (the reference processes channel 0 and
optionally channel 1 separately: friture/spectrum.py:149-153, friture/delay_estimator.py:97-98),
so the channel axis is block-partitioned across ranks and the data path needs no collective.
Two collectives exist around it, both tiny:

  * plan time: rank 0 broadcasts the constant tables (weighting curve, colour LUT, filter
    coefficients) so that every rank computes with bit-identical constants;
  * after a batch: per-channel summaries (band energies, spectrogram digests) are all-gathered so
    that rank 0 can present all channels.  Full PSD / pixel slabs stay resident on the GPU that
    produced them by default (an all-gather of those is per-link bound on xGMI: 7 links x ~153 GB/s
    per GPU); `SlabGather` is the OPTIONAL gather of SURVEY.md §8e for a consumer that wants every
    channel's slab on every GPU: one all-gather per batch, issued asynchronously behind the batch
    that produced the slab and overlapped with the next batch's kernels (double-buffered).

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


def _solo() -> bool:
    """True when no collective needs to run: no process group, or a group of one.  FRT_DIST_FORCE=1 makes a group of one go
    through the collective library anyway (tests/test_rccl_single_rank_gpu.py: every call of this module on RCCL with device
    tensors on the one GPU a test box has)."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return True
    return dist.get_world_size() == 1 and os.environ.get("FRT_DIST_FORCE", "") != "1"


def world_is_one() -> bool:
    """True when the job is ONE process and nothing forces the collective library (bench.py's timed region then brackets its steps
    with synchronize alone; with FRT_DIST_FORCE=1 the group of one still goes through RCCL, barriers included)."""
    return _solo()


def init_process_group(backend: str | None = None, device=None):
    """Join the job's process group (RCCL on GPU, gloo on CPU).  No-op for single-process runs (unless FRT_DIST_FORCE=1)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if (world == 1 and os.environ.get("FRT_DIST_FORCE", "") != "1") or dist.is_initialized():
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
    if _solo():
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
    if _solo():
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


def gather_scalars(value: float, device=None) -> list:
    """One float from every rank, in rank order ([value] for a single process): per-rank step times next to the maximum."""
    import torch
    import torch.distributed as dist
    if _solo():
        return [float(value)]
    mine = torch.tensor([value], dtype=torch.float64, device=device)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [float(p.item()) for p in parts]


class SlabGather:
    """Optional all-gather of the per-rank output slabs, overlapped with the next batch.

    `start(slab, slot)` enqueues an asynchronous all-gather of this rank's slab [C_local, ...] into receive buffer `slot`
    ([world, C_local, ...]); with RCCL the collective is ordered behind the kernels already enqueued on the current stream
    (the ones that produce the slab) and runs on the communicator's own stream, so the kernels of the next batch, enqueued
    afterwards, overlap with it.  `wait(slot)` must be called before the slab buffer that was gathered is overwritten and
    before the receive buffer is read.  Every rank must pass slabs of the same shape (equal channel counts).  A single
    process needs no communication: the "gathered" tensor is a view of the slab."""

    def __init__(self, like, n_slots: int = 2):
        import torch
        import torch.distributed as dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.solo = _solo()
        self.recv = [torch.empty((self.world,) + tuple(like.shape), dtype=like.dtype, device=like.device) for _ in range(n_slots)] \
            if not self.solo else [None] * n_slots
        self.work = [None] * n_slots
        self.bytes_per_gather = self.world * like.numel() * like.element_size()

    def start(self, slab, slot: int):
        import torch.distributed as dist
        self.wait(slot)
        if self.solo:
            self.recv[slot] = slab[None]
            return
        self.work[slot] = dist.all_gather_into_tensor(self.recv[slot].view(-1), slab.contiguous().view(-1), async_op=True)

    def wait(self, slot: int):
        w = self.work[slot]
        if w is not None:
            w.wait()
            self.work[slot] = None
        return self.recv[slot]

    def wait_all(self):
        for s in range(len(self.work)):
            self.wait(s)


def gather_ranks(device=None) -> list:
    """The rank ids an all-gather over the job's process group returns ([0] for a single process): evidence that the
    collective library (RCCL on GPUs) saw every rank."""
    import torch
    import torch.distributed as dist
    if _solo():
        return [0]
    mine = torch.tensor([dist.get_rank()], dtype=torch.int64, device=device)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [int(p.item()) for p in parts]


def max_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if _solo():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(device=None):
    import torch.distributed as dist
    if not _solo():
        if device is not None and getattr(device, "type", "cpu") == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()
