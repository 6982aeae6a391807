"""CPU tests (no GPU) of the node-shared gather-v that carries the chunks' run records to rank 0's host merge
(ribbit_amd/node_gather.py), with the rendezvous done over gloo exactly as bench.py does it, world_size 2 and 3."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import ribbit_amd
from ribbit_amd.distributed import open_node_gather
from ribbit_amd.node_gather import NodeGather

STEPS = 7


def _records(rank, step, n):
    rs = np.random.RandomState(50_000 + 1000 * rank + step)
    out = np.zeros(n, ribbit_amd.RUN_DT)
    out["start"] = rs.randint(0, 1 << 30, n)
    out["end"] = out["start"] + rs.randint(1, 1000, n)
    out["mlen"] = rs.randint(2, 101, n)
    out["term"] = rs.randint(-1, 3, n)
    return out


def _worker(rank, world, port, cap, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ng = open_node_gather(ribbit_amd.RUN_DT, cap, 16)
        ok = True
        for k in range(1, STEPS + 1):
            if rank == 0 and k > 1:
                ng.release(k - 1)
            ng.wait_free(k)
            rec, hv = ng.mine(k)
            n = (37 * rank + 11 * k) % cap
            rec[:n] = _records(rank, k, n)
            hv[:rank] = _records(rank, -k, rank)
            ng.publish(k, n, rank)
            if rank == 0:
                parts, halves = ng.collect(k)
                for r in range(world):
                    nr = (37 * r + 11 * k) % cap
                    ok &= np.array_equal(parts[r], _records(r, k, nr)) and np.array_equal(halves[r], _records(r, -k, r))
        dist.barrier()
        ng.close()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_node_gather_delivers_every_ranks_records_step_after_step(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 500, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == {r: True for r in range(world)}


def test_producer_cannot_run_more_than_nslots_steps_ahead():
    ng = NodeGather(ribbit_amd.RUN_DT, 8, 2, rank=0, world=1)
    try:
        ng.wait_free(1); ng.publish(1, 0, 0)
        ng.wait_free(2); ng.publish(2, 0, 0)
        with pytest.raises(TimeoutError):
            ng.wait_free(3, timeout=0.05)          # step 1 (same slot) has not been released
        ng.collect(1); ng.release(1)
        ng.wait_free(3)
    finally:
        ng.close()


def test_join_run_halves_pairs_by_motif_and_position_and_rejects_garbage():
    dt = ribbit_amd.RUN_DT
    hs, he = ribbit_amd.RUN_HALF_START, ribbit_amd.RUN_HALF_END
    a = np.array([(100, -1, 5, hs), (7000, -1, 5, hs), (50, -1, 9, hs)], dtype=dt)
    b = np.array([(-1, 900, 5, he + 0), (-1, 300, 9, he + 2), (-1, 7500, 5, he + 1)], dtype=dt)
    got = ribbit_amd.join_run_halves([a, b])
    want = np.array([(100, 900, 5, 0), (7000, 7500, 5, 1), (50, 300, 9, 2)], dtype=dt)
    assert np.array_equal(got, want)
    assert len(ribbit_amd.join_run_halves([])) == 0
    with pytest.raises(ribbit_amd.RibbitHipError):
        ribbit_amd.join_run_halves([a[:2], b])                       # a start is missing
    with pytest.raises(ribbit_amd.RibbitHipError):
        ribbit_amd.join_run_halves([a, np.array([(-1, 90, 5, he), (-1, 300, 9, he), (-1, 7500, 5, he)], dtype=dt)])   # end before start
    whole = np.array([(1, 20, 3, 0), (0, 0, 0, ribbit_amd.RUN_NOT_OWNED), (500, 600, 2, 1)], dtype=dt)
    merged = ribbit_amd.merge_chunk_runs([whole], [a, b])
    assert list(merged["mlen"]) == [2, 3, 5, 5, 9] and list(merged["start"]) == [500, 1, 100, 7000, 50]


def test_a_segment_that_does_not_fit_fails_at_creation_and_leaves_nothing_behind():
    before = set(os.listdir("/dev/shm"))
    with pytest.raises(OSError):
        NodeGather(ribbit_amd.RUN_DT, 10 ** 13, 2, rank=0, world=1)          # 160 TB
    assert set(os.listdir("/dev/shm")) == before
