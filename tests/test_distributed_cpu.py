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


def test_bench_rank_logic_two_ranks_gloo():
    """bench.py end to end with WORLD_SIZE = 2 over gloo and its CPU stand-in engine (--stub-engine): channel sharding,
    table broadcast from rank 0, barrier-bracketed timing with max over ranks, digest all-gather, ranks_seen — the
    contract's launch line, one JSON line from rank 0."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--stub-engine", "--log2-samples", "14", "--batches", "2", "--cpu-budget", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=str(root))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["ranks_seen"] == [0, 1] and rec["data"] == "stub"
    F = (2 ** 14 - 1024) // 512 + 1
    assert rec["config"]["channels"] == 2 and rec["config"]["spectra_per_step"] == 2 * F
    # whole-job value = spectra of all ranks / max-over-ranks time
    assert abs(rec["value"] - 2 * F * 3 / (rec["ms_per_step"] * 3e-3)) <= 1e-6 * rec["value"]
    # digest: both ranks' channels arrived (the stub writes 1000 * first sample of each channel)
    import numpy as np
    want = sum(float(np.int32(1000 * (0.25 * np.random.default_rng(42 + c).standard_normal(2 ** 14, dtype=np.float32))[0]))
               for c in range(2))
    assert abs(rec["digest"] - want) < 1e-6
    # the CPU baseline is emitted by rank 0 at every world size: spectrogram, both banks, GCC-PHAT —
    # one core and every core, each naming its sample
    cb = rec["cpu_baseline"]
    assert cb["cores"] == 1 and cb["kind"] == "port" and cb["value"] > 0 and cb["all_cores"]["cores"] >= 1 and cb["all_cores"]["value"] > 0
    side = rec["cpu_baseline_legs"]
    assert set(side) == {"octave_iir_bpo3", "octave_ola_bpo3", "octave_iir_bpo24", "octave_ola_bpo24", "gcc_phat"}
    for v in side.values():
        assert v["value"] > 0 and v["cores"] == 1 and v["all_cores"]["value"] > 0 and v["sample"]


def test_bench_gpus_2_starts_itself_without_a_launcher():
    """plain `python bench.py --gpus 2` (no torchrun, WORLD_SIZE unset) re-executes itself under
    torch.distributed.run — one JSON line from rank 0 with both ranks seen, the same record as the contract's launch line."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--stub-engine", "--log2-samples", "14", "--batches", "2", "--cpu-budget", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=str(root), env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == [0, 1] and rec["steps"] == 3 and rec["warmup"] == 1
    F = (2 ** 14 - 1024) // 512 + 1
    assert rec["config"]["channels"] == 2 and rec["config"]["spectra_per_step"] == 2 * F


def test_bench_two_ranks_gloo_with_slab_gather():
    """The optional slab all-gather of SURVEY.md §8e (--gather-slabs): issued asynchronously behind every step, one receive
    buffer per rotating batch, verified against every rank's own slab; per-rank step times are reported next to the
    maximum; ranks_seen is asserted by the job itself."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(root / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--stub-engine", "--log2-samples", "14", "--batches", "2", "--gather-slabs", "--cpu-budget", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=str(root))
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    F = (2 ** 14 - 1024) // 512 + 1
    sg = rec["slab_gather"]
    assert sg["enabled"] and sg["slabs_verified"] is True and sg["bytes_per_step_per_rank"] == 2 * F * 513 * 4
    pr = rec["per_rank"]
    assert len(pr["ms_per_step"]) == 2 and pr["ms_per_step_min"] <= pr["ms_per_step_max"]
    assert abs(pr["ms_per_step_max"] - rec["ms_per_step"]) <= 1e-4 * rec["ms_per_step"]      # (per-rank figures are printed at 5 digits)


def _slab_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    distributed.init_process_group(backend="gloo")
    slabs = [torch.full((3, 4, 5), float(10 * b + rank)) for b in range(2)]
    g = distributed.SlabGather(slabs[0], n_slots=2)
    for step in range(5):                                   # double-buffered: start(b) waits for the previous gather of slot b
        b = step % 2
        slabs[b] += 100.0                                   # "the next batch" overwrites the buffer only after wait(b) inside start
        g.wait(b)
        g.start(slabs[b], b)
    g.wait_all()
    for b in range(2):
        got = g.wait(b)
        for r in range(world):
            n_updates = 3 if b == 0 else 2
            assert torch.all(got[r] == 10 * b + r + 100.0 * n_updates), (b, r, got[r].flatten()[0])
    assert distributed.gather_scalars(0.5 + rank) == [0.5 + r for r in range(world)]
    dist.destroy_process_group()
    q.put(rank)


def test_slab_gather_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_slab_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]


def test_bench_stub_single_process():
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--stub-engine", "--log2-samples", "13", "--steps", "2", "--warmup", "1", "--cpu-budget", "0"],
                       capture_output=True, text=True, timeout=300, cwd=str(root))
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["ranks_seen"] == [0] and rec["roofline"]["traffic"] is None


def _forced_solo_worker(port, q):
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FRT_DIST_FORCE="1")
    import torch
    import torch.distributed as dist
    r, lr, w = distributed.init_process_group(backend="gloo")
    assert (r, w) == (0, 1) and dist.is_initialized()          # a group of one exists only because it was forced
    tabs = {"weight": np.linspace(-3, 1, 33), "lut": np.arange(256, dtype=np.uint32) | 0xFF000000}
    got = distributed.broadcast_tables(tabs)
    assert np.array_equal(got["weight"], tabs["weight"]) and got["lut"].dtype == np.uint32 and np.array_equal(got["lut"], tabs["lut"])
    local = torch.arange(6, dtype=torch.float64).reshape(3, 2)
    assert torch.equal(distributed.gather_channel_summaries(local, 3), local)
    assert distributed.gather_scalars(0.25) == [0.25] and distributed.gather_ranks() == [0] and distributed.max_over_ranks(2.5) == 2.5
    slab = torch.arange(24, dtype=torch.float32).reshape(2, 3, 4)
    sg = distributed.SlabGather(slab, n_slots=2)
    assert not sg.solo                                          # the collective really runs
    sg.start(slab, 1)
    g = sg.wait(1)
    assert tuple(g.shape) == (1, 2, 3, 4) and torch.equal(g[0], slab)
    distributed.barrier()
    dist.destroy_process_group()
    q.put("ok")


def test_forced_group_of_one_runs_the_collectives():
    """FRT_DIST_FORCE=1 (what tests/test_rccl_single_rank_gpu.py uses on the GPU box for RCCL): with a world of one the helpers
    normally short-cut; forced, they go through the collective library — here gloo."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_solo_worker, args=(_free_port(), q))
    p.start()
    p.join(120)
    assert p.exitcode == 0 and q.get(timeout=5) == "ok"
