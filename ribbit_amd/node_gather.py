"""node_gather -- gather-v of per-rank record arrays into host memory shared by the ranks of ONE node.

Why not RCCL for this step: the consumer of the candidate intervals is the host-side merge on rank 0
(merge_types.cpp / parse_perfect_shiftxor.cpp:46-143 are sequential), so the bytes have to end up in host memory
anyway.  Gathering them on rank 0's GPU over xGMI and copying them down rank 0's single PCIe link serialises N
chunks' records on one link; letting every GPU copy its own chunk's records down its OWN PCIe link, straight into
a page-locked segment all ranks map, uses N links in parallel and needs no second hop.  Across nodes (or when the
segment cannot be created) ribbit_amd.distributed.gather_array over RCCL is the transport.

Layout of the segment (one per job):  nslots x world x { header line | records[cap] | halves[half_cap] }, and
one acknowledgement line.  Slots alternate between steps, so rank r may fill slot (k+1) % nslots while rank 0
still reads slot k % nslots.  Protocol for step k = 1, 2, ...:
  every rank:  wait_free(k) -> fill mine(k) in place -> publish(k, n, n_halves)
  rank 0:      collect(k) -> views of every rank's records -> ... -> release(k)
Ordering relies on x86-64 stores being observed in program order (count/records are written before the
sequence number that announces them; the DMA that filled the records has completed before publish()).
"""
from __future__ import annotations

import mmap
import os
import time
import uuid

import numpy as np

_LINE = 64            # bytes; header and acknowledgement each own a cache line
_H_SEQ, _H_N, _H_NHALF = 0, 1, 2


class NodeGather:
    def __init__(self, dtype: np.dtype, cap: int, half_cap: int, rank: int, world: int, name: str | None = None,
                 nslots: int = 2):
        """Rank 0 creates the segment (name=None) and hands `self.name` to the other ranks, which attach."""
        self.dtype, self.cap, self.half_cap = np.dtype(dtype), int(cap), int(half_cap)
        self.rank, self.world, self.nslots = rank, world, nslots
        self._cell = _LINE + (self.cap + self.half_cap) * self.dtype.itemsize
        self._cell = (self._cell + 4095) // 4096 * 4096          # page aligned cells: each rank page-locks only its own
        self.nbytes = 4096 + nslots * world * self._cell
        # a plain file in /dev/shm mapped MAP_SHARED (multiprocessing.shared_memory would have every attaching
        # process's resource tracker unlink the segment when that process exits)
        self._owner = name is None
        self.name = name if name is not None else f"/dev/shm/ribbit_gather_{os.getpid()}_{uuid.uuid4().hex[:12]}"
        fd = os.open(self.name, (os.O_CREAT | os.O_EXCL | os.O_RDWR) if self._owner else os.O_RDWR, 0o600)
        try:
            if self._owner:
                # reserve the pages now (zero-filled: no step published, nothing released): a tmpfs that is too small
                # must fail here with ENOSPC, not later with SIGBUS on first touch
                try:
                    os.posix_fallocate(fd, 0, self.nbytes)
                except OSError:
                    os.unlink(self.name)
                    raise
            elif os.fstat(fd).st_size != self.nbytes:
                raise ValueError(f"{self.name}: segment size differs from this rank's cap/half_cap/world")
            self._map = mmap.mmap(fd, self.nbytes, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
        finally:
            os.close(fd)
        self._ack = np.frombuffer(self._map, dtype=np.int64, count=8, offset=0)

    # ---- layout
    def _offset(self, slot: int, rank: int) -> int:
        return 4096 + (slot * self.world + rank) * self._cell

    def _header(self, slot: int, rank: int) -> np.ndarray:
        return np.frombuffer(self._map, dtype=np.int64, count=_LINE // 8, offset=self._offset(slot, rank))

    def _records(self, slot: int, rank: int) -> np.ndarray:
        return np.frombuffer(self._map, dtype=self.dtype, count=self.cap, offset=self._offset(slot, rank) + _LINE)

    def _halves(self, slot: int, rank: int) -> np.ndarray:
        return np.frombuffer(self._map, dtype=self.dtype, count=self.half_cap,
                             offset=self._offset(slot, rank) + _LINE + self.cap * self.dtype.itemsize)

    def my_cells(self):
        """(address, bytes) of this rank's cell in every slot, for page-locking (ribbit_hip_host_register)."""
        base = np.frombuffer(self._map, dtype=np.uint8).ctypes.data
        return [(base + self._offset(s, self.rank), self._cell) for s in range(self.nslots)]

    # ---- producer side (every rank)
    def wait_free(self, step: int, timeout: float = 60.0) -> None:
        """Block until rank 0 has released the step that last used this step's slot."""
        need = step - self.nslots
        t0 = time.perf_counter()
        while self._ack[0] < need:
            if time.perf_counter() - t0 > timeout:
                raise TimeoutError(f"rank {self.rank}: step {need} was never released by rank 0")

    def mine(self, step: int):
        """-> (records array, halves array) of this rank for `step`, to be filled in place."""
        s = step % self.nslots
        return self._records(s, self.rank), self._halves(s, self.rank)

    def publish(self, step: int, n: int, n_halves: int) -> None:
        h = self._header(step % self.nslots, self.rank)
        h[_H_N] = n
        h[_H_NHALF] = n_halves
        h[_H_SEQ] = step

    # ---- consumer side (rank 0)
    def collect(self, step: int, timeout: float = 60.0):
        """Wait for every rank's step `step`; -> (list of record views, list of halves views), in rank order."""
        s = step % self.nslots
        parts, halves = [], []
        t0 = time.perf_counter()
        for r in range(self.world):
            h = self._header(s, r)
            while h[_H_SEQ] != step:
                if time.perf_counter() - t0 > timeout:
                    raise TimeoutError(f"rank {r} never published step {step}")
            parts.append(self._records(s, r)[:int(h[_H_N])])
            halves.append(self._halves(s, r)[:int(h[_H_NHALF])])
        return parts, halves

    def release(self, step: int) -> None:
        self._ack[0] = step

    def close(self) -> None:
        self._ack = None
        try:
            self._map.close()
        except BufferError:
            pass                      # views handed out by mine()/collect() are still alive; the mapping goes with them
        if self._owner:
            try:
                os.unlink(self.name)
            except FileNotFoundError:
                pass
