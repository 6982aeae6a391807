"""Exchange step of the multi-GPU path: a gather-v of sparse 16-byte records (runs, calls or seeds) across
ranks -- one process per GPU, torch.distributed with backend "nccl" (= RCCL over xGMI on ROCm) on GPUs or "gloo" in the
CPU tests.  The scan itself shards by record with no collective; this is the only communication on the path
(BASELINE.json north_star: "RCCL all-gatherv of candidate seed intervals before host-side merge").  The records start
in HBM (the pairing kernels write them there) and DeviceGather moves them GPU-to-GPU: grouped send / recv to the rank
that runs the host merge, nothing staged through numpy.  Payloads are KB-MB, i.e. latency-bound: one small count
all-gather plus one message per rank, nothing ring-shaped to tune.  The host-array helpers below serve the CPU tests
(gloo) and the one-off exchanges outside the timed path (halos, verification)."""
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


class _DeviceView:
    """zero-copy view of `nbytes` of device memory for torch.as_tensor (the CUDA array interface, which ROCm builds of
    torch read the same way)"""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def device_bytes(ptr: int, nbytes: int, device: torch.device) -> torch.Tensor:
    """uint8 tensor aliasing device memory the C ABI owns (no copy)"""
    try:
        if nbytes == 0 or not ptr:
            return torch.empty(0, dtype=torch.uint8, device=device)
        return torch.as_tensor(_DeviceView(ptr, nbytes), device=device)
    except RuntimeError as e:
        import ribbit_amd
        if ribbit_amd.loaded_before_torch and "No HIP GPUs" in str(e):
            raise RuntimeError("torch cannot use the GPU in this process because libribbit_hip.so was loaded before torch was imported "
                               "(two HIP runtimes, tests/conftest.py): import torch first") from e
        raise


class DeviceGather:
    """gather-v of device-resident 16-byte records to rank `dst` over RCCL: the candidate seed intervals every rank's
    pairing kernels left in HBM travel GPU-to-GPU over xGMI (grouped send / recv, one message per rank, no padding) into
    one buffer on dst's GPU and come down dst's PCIe link once, into page-locked memory, for the host merge.  Only
    counts cross in a collective of their own (one small all-gather).  Buffers are reused from step to step."""

    def __init__(self, device: torch.device, dst: int = 0):
        self.dev, self.dst = device, dst
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.counts = torch.zeros(2, dtype=torch.int64, device=device)
        self.recv = torch.empty(0, dtype=torch.uint8, device=device)
        self.host = torch.empty(0, dtype=torch.uint8).pin_memory() if device.type == "cuda" else torch.empty(0, dtype=torch.uint8)

    def _room(self, nbytes: int):
        if self.recv.numel() < nbytes:
            self.recv = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=self.dev)
        if self.host.numel() < nbytes:
            self.host = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8)
            if self.dev.type == "cuda":
                self.host = self.host.pin_memory()

    def gather(self, runs, n_runs: int, halves, n_halves: int, dtype):
        """runs / halves: device pointers (int) of n_runs / n_halves 16-byte records, or uint8 tensors on self.dev holding
        them.  -> on dst: ([runs of rank 0, runs of rank 1, ...], [halves ...]) as numpy views of page-locked memory that
        stay valid until the next call; elsewhere (None, None).

        Both ends post their point-to-point operations through ONE mechanism, dist.batch_isend_irecv (a coalesced group on
        the process group's own communicator): a plain dist.send on one side against a batched irecv on the other would use
        two different RCCL communicators and never meet.  Nothing is copied on the sending side: the records are sent from
        where the pairing kernels wrote them (two messages, runs then halves, matched in order), and the stream is fenced
        before the call returns so that the handle may reuse its buffer."""
        rec = 16
        as_bytes = lambda x, n: x[:n * rec] if isinstance(x, torch.Tensor) else device_bytes(x, n * rec, self.dev)
        self.counts[0], self.counts[1] = n_runs, n_halves
        gathered = [torch.zeros_like(self.counts) for _ in range(self.world)]
        dist.all_gather(gathered, self.counts)            # also creates the world communicator the grouped operations use
        counts = torch.stack(gathered).cpu().numpy()
        sizes = [(int(c[0]) + int(c[1])) * rec for c in counts]
        ops = []
        total, offs = 0, None
        if self.rank != self.dst:
            # the order of a pair's messages is the contract: runs first, then halves; empty messages are posted by neither side
            if n_runs:
                ops.append(dist.P2POp(dist.isend, as_bytes(runs, n_runs), self.dst))
            if n_halves:
                ops.append(dist.P2POp(dist.isend, as_bytes(halves, n_halves), self.dst))
        else:
            total = sum(sizes)
            self._room(total)
            offs = np.concatenate(([0], np.cumsum(sizes)))
            for r in range(self.world):
                a, nr, nh = int(offs[r]), int(counts[r][0]) * rec, int(counts[r][1]) * rec
                if r == self.dst:
                    if nr:
                        self.recv[a:a + nr].copy_(as_bytes(runs, n_runs))
                    if nh:
                        self.recv[a + nr:a + nr + nh].copy_(as_bytes(halves, n_halves))
                    continue
                if nr:
                    ops.append(dist.P2POp(dist.irecv, self.recv[a:a + nr], r))
                if nh:
                    ops.append(dist.P2POp(dist.irecv, self.recv[a + nr:a + nr + nh], r))
        self.last_ops = [(op.op.__name__, int(op.peer)) for op in ops]      # what this rank posted (tests)
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        if self.rank == self.dst:
            self.host[:total].copy_(self.recv[:total], non_blocking=True)
        if self.dev.type == "cuda":
            torch.cuda.current_stream(self.dev).synchronize()      # sends have read the handle's buffer; the records are on dst's host
        if self.rank != self.dst:
            return None, None
        flat = self.host[:total].numpy()
        out_runs, out_halves = [], []
        for r in range(self.world):
            a, nr, nh = int(offs[r]), int(counts[r][0]) * rec, int(counts[r][1]) * rec
            out_runs.append(flat[a:a + nr].view(dtype))
            out_halves.append(flat[a + nr:a + nr + nh].view(dtype))
        return out_runs, out_halves


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
