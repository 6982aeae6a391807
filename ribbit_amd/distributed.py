"""Exchange step of the multi-GPU path: an all-gather-v of sparse 16-byte records (runs, calls or
seeds) across ranks -- one process per GPU, torch.distributed with backend "nccl" (= RCCL over xGMI on
ROCm) on GPUs or "gloo" in the CPU tests.  The scan itself shards by record with no collective; this is
the only communication on the path (BASELINE.json north_star: "RCCL all-gatherv of candidate seed
intervals before host-side merge").  Payloads are KB-MB, i.e. latency-bound: one count all-gather plus
one padded all-gather, nothing ring-shaped to tune."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def allgather_records(records: np.ndarray, device: torch.device | None = None) -> list[np.ndarray]:
    """Gather a structured array of 16-byte records (four int32 fields) from every rank.

    Returns one array per rank (same dtype), identical on all ranks.  `device` selects where the
    staging tensors live: the rank's GPU for nccl, CPU (None) for gloo.
    """
    assert records.dtype.itemsize == 16, "records are four int32 fields"
    world = dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([len(records)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    buf = torch.zeros((cap, 4), dtype=torch.int32, device=dev)
    if len(records):
        flat = np.ascontiguousarray(records).view("<i4").reshape(-1, 4)
        buf[:len(records)] = torch.from_numpy(flat).to(dev)
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf)
    out = []
    for r in range(world):
        arr = gathered[r][:counts[r]].cpu().numpy()
        out.append(np.ascontiguousarray(arr).view(records.dtype).reshape(-1))
    return out


def allgather_array(arr: np.ndarray, device: torch.device | None = None) -> list[np.ndarray]:
    """all-gather-v of a 1-D numpy array of any fixed-size dtype (count exchange + padded gather)."""
    world = dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
    n = torch.tensor([raw.size], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    buf = torch.zeros(max(max(counts), 1), dtype=torch.uint8, device=dev)
    if raw.size:
        buf[:raw.size] = torch.from_numpy(raw).to(dev)
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf)
    return [gathered[r][:counts[r]].cpu().numpy().view(arr.dtype).copy() for r in range(world)]


def gather_array(arr: np.ndarray, device: torch.device | None = None, dst: int = 0):
    """gather-v of a 1-D numpy array to rank `dst` (count exchange + padded gather).  Returns the list of
    per-rank arrays on `dst`, None elsewhere.  Only `dst` pays for receiving world x payload."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
    n = torch.tensor([raw.size], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    buf = torch.zeros(max(max(counts), 1), dtype=torch.uint8, device=dev)
    if raw.size:
        buf[:raw.size] = torch.from_numpy(raw).to(dev)
    gathered = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, gathered, dst=dst)
    if rank != dst:
        return None
    return [gathered[r][:counts[r]].cpu().numpy().view(arr.dtype).copy() for r in range(world)]


def same_node() -> bool:
    """True when every rank of the job runs on this node (torchrun sets LOCAL_WORLD_SIZE)."""
    import os
    return int(os.environ.get("LOCAL_WORLD_SIZE", "0")) == dist.get_world_size()


def open_node_gather(dtype, cap: int, half_cap: int, nslots: int = 2):
    """Collective: rank 0 creates the node-shared segment (ribbit_amd.node_gather), its name is broadcast and
    the other ranks attach.  cap / half_cap must be the same on all ranks.  Returns None on every rank when the
    segment cannot be created or attached (e.g. /dev/shm too small): the caller then gathers over RCCL."""
    from .node_gather import NodeGather
    rank, world = dist.get_rank(), dist.get_world_size()
    ng = None
    if rank == 0:
        try:
            ng = NodeGather(dtype, cap, half_cap, rank, world, nslots=nslots)
        except OSError:
            ng = None
    box = [ng.name if ng is not None else None]
    dist.broadcast_object_list(box, src=0)
    ok = 1
    if rank != 0 and box[0] is not None:
        try:
            ng = NodeGather(dtype, cap, half_cap, rank, world, name=box[0], nslots=nslots)
        except (OSError, ValueError):
            ok = 0
    if box[0] is None:
        ok = 0
    flags = [None] * world
    dist.all_gather_object(flags, ok)
    if not all(flags):
        if ng is not None:
            ng.close()
        return None
    return ng


def shard_records(n_records: int, lengths: list[int], world: int) -> list[list[int]]:
    """Longest-first bin packing of record indices over ranks (option 1)."""
    order = sorted(range(n_records), key=lambda i: -lengths[i])
    load = [0] * world
    bins: list[list[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: load[k])
        bins[r].append(i)
        load[r] += lengths[i]
    return bins
