"""The N > 1 path on CPU: world_size-2 gloo run of the record-sharded exchange (no GPU compute:
each rank replays call lists through the product's host merges and the seed lists are all-gathered)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch.distributed as dist

    import ribbit_amd
    from cases import simulated_cases
    from oracle_lib import LIST_PERFECT, LIST_SUBST, Oracle
    from ribbit_amd.distributed import allgather_records, shard_records

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cases = simulated_cases()[:3]
    bins = shard_records(len(cases), [len(c[1]) for c in cases], world)
    mine = []
    for idx in bins[rank]:
        name, seq, m_lo, m_hi = cases[idx]
        with Oracle(seq, m_lo, m_hi) as o:           # stands in for the GPU scan: provides the call lists
            o.run_perfect(); o.run_subst()
            r = ribbit_amd.host_replay_calls(m_lo, m_hi, seq, o.calls(LIST_PERFECT), o.calls(LIST_SUBST))
        seeds = r["subst"].copy()
        tagged = np.zeros(len(seeds), dtype=ribbit_amd.SEED_DT)
        tagged[:] = seeds
        tagged["type"] = idx                          # record index travels in the spare field
        mine.append(tagged)
    local = np.concatenate(mine) if mine else np.zeros(0, ribbit_amd.SEED_DT)
    parts = allgather_records(local)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.concatenate(parts))
    dist.barrier()
    dist.destroy_process_group()


def test_record_sharded_exchange_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "rank0.npy")
    b = np.load(tmp_path / "rank1.npy")
    assert np.array_equal(a, b), "all ranks must hold the same gathered records"

    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import ribbit_amd
    from cases import simulated_cases
    from oracle_lib import LIST_SUBST, Oracle
    for idx, (name, seq, m_lo, m_hi) in enumerate(simulated_cases()[:3]):
        with Oracle(seq, m_lo, m_hi) as o:
            o.run_perfect(); o.run_subst()
            want = o.seeds(LIST_SUBST)
        got = a[a["type"] == idx]
        assert np.array_equal(got["start"], want["start"]) and np.array_equal(got["end"], want["end"])
        assert np.array_equal(got["mlen"], want["mlen"])


def test_shard_records_balances_longest_first():
    from ribbit_amd.distributed import shard_records
    bins = shard_records(5, [10, 50, 20, 40, 30], 2)
    assert sorted(sum(bins, [])) == [0, 1, 2, 3, 4]
    loads = [sum([10, 50, 20, 40, 30][i] for i in b) for b in bins]
    assert abs(loads[0] - loads[1]) <= 10


def _gather_worker(rank, world, port, out_dir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch
    import torch.distributed as dist

    import ribbit_amd
    from ribbit_amd.distributed import DeviceGather

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dg = DeviceGather(torch.device("cpu"))
    ok = True
    for step in range(4):                            # buffers are reused and must grow
        rs = np.random.RandomState(100 * step + rank)
        n, nh = int(rs.randint(0, 5000 * (step + 1))) if not (step == 1 and rank == 1) else 0, int(rs.randint(0, 40))
        runs = np.zeros(n, ribbit_amd.RUN_DT)
        runs["start"] = rs.randint(0, 1 << 30, n); runs["end"] = runs["start"] + 5; runs["mlen"] = rank; runs["term"] = step
        halves = np.zeros(nh, ribbit_amd.RUN_DT)
        halves["start"] = rs.randint(0, 1 << 30, nh); halves["mlen"] = rank; halves["term"] = 3
        tr = torch.from_numpy(runs.view(np.uint8).reshape(-1).copy())
        th = torch.from_numpy(halves.view(np.uint8).reshape(-1).copy())
        got_runs, got_halves = dg.gather(tr, n, th, nh, ribbit_amd.RUN_DT)
        if rank == 0:
            for r in range(world):
                rs2 = np.random.RandomState(100 * step + r)
                n2 = int(rs2.randint(0, 5000 * (step + 1))) if not (step == 1 and r == 1) else 0
                nh2 = int(rs2.randint(0, 40))
                want_start = rs2.randint(0, 1 << 30, n2)
                want_hstart = rs2.randint(0, 1 << 30, nh2)
                ok &= len(got_runs[r]) == n2 and len(got_halves[r]) == nh2
                ok &= bool(np.array_equal(got_runs[r]["start"], want_start) and np.all(got_runs[r]["mlen"] == r) and np.all(got_runs[r]["term"] == step))
                ok &= bool(np.array_equal(got_halves[r]["start"], want_hstart) and np.all(got_halves[r]["term"] == 3))
        else:
            ok &= got_runs is None and got_halves is None
    with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
        f.write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def test_device_gather_delivers_every_ranks_records_to_rank0(tmp_path):
    """DeviceGather (the RCCL path of the N > 1 bench: count all-gather + one message per rank into one buffer on rank 0)
    with CPU tensors over gloo, world size 3: variable sizes incl. an empty rank, buffers that grow between steps."""
    port = _free_port()
    mp.spawn(_gather_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    assert all(open(tmp_path / f"ok{r}").read() == "1" for r in range(3))


def _gather4_worker(rank, world, port, out_dir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch
    import torch.distributed as dist

    import ribbit_amd
    from ribbit_amd.distributed import DeviceGather

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # every point-to-point operation of the exchange must go through the batched API on BOTH ends (over RCCL a plain
    # send would open a pair communicator that the receiver's grouped irecv never joins): count what is called
    calls = {"batch": 0, "plain": 0}
    real_batch, real_send, real_recv = dist.batch_isend_irecv, dist.send, dist.recv

    def counting_batch(ops):
        calls["batch"] += 1
        return real_batch(ops)

    def plain(*a, **k):
        calls["plain"] += 1
        raise AssertionError("plain point-to-point call in DeviceGather")

    dist.batch_isend_irecv, dist.send, dist.recv = counting_batch, plain, plain
    dg = DeviceGather(torch.device("cpu"))
    ok = True
    posted = []

    def sizes(step, r):
        # ranks 1 and 3 are empty in steps 0 and 2, rank 2 in step 1, everybody in step 4; the counts change every step
        if (r in (1, 3) and step in (0, 2)) or (r == 2 and step == 1) or step == 4:
            return 0, 0
        return 1000 * (step + 1) + 37 * r, (step + r) % 3

    for step in range(6):
        n, nh = sizes(step, rank)
        runs = np.zeros(n, ribbit_amd.RUN_DT)
        runs["start"] = np.arange(n) + 1000 * rank; runs["mlen"] = rank; runs["term"] = step
        halves = np.zeros(nh, ribbit_amd.RUN_DT)
        halves["end"] = np.arange(nh) + 7 * step; halves["mlen"] = rank; halves["term"] = 4
        got_runs, got_halves = dg.gather(torch.from_numpy(runs.view(np.uint8).reshape(-1).copy()), n,
                                         torch.from_numpy(halves.view(np.uint8).reshape(-1).copy()), nh, ribbit_amd.RUN_DT)
        posted.append(list(dg.last_ops))
        if rank == 0:
            for r in range(world):
                n2, nh2 = sizes(step, r)
                ok &= len(got_runs[r]) == n2 and len(got_halves[r]) == nh2
                ok &= bool(np.array_equal(got_runs[r]["start"], np.arange(n2) + 1000 * r) and np.all(got_runs[r]["term"] == step))
                ok &= bool(np.array_equal(got_halves[r]["end"], np.arange(nh2) + 7 * step) and np.all(got_halves[r]["mlen"] == r))
        else:
            ok &= got_runs is None and got_halves is None
            # a sender posts exactly its non-empty messages, as isend to rank 0
            want = [("isend", 0)] * ((n > 0) + (nh > 0))
            ok &= posted[-1] == want
    ok &= calls["plain"] == 0
    if rank == 0:
        # rank 0 posts one irecv per non-empty message of every other rank, in rank order
        for step in range(6):
            want = []
            for r in range(1, world):
                n2, nh2 = sizes(step, r)
                want += [("irecv", r)] * ((n2 > 0) + (nh2 > 0))
            ok &= posted[step] == want
    with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
        f.write("1" if ok else "0")
    dist.batch_isend_irecv, dist.send, dist.recv = real_batch, real_send, real_recv
    dist.barrier()
    dist.destroy_process_group()


def test_device_gather_world4_with_empty_ranks_and_one_p2p_mechanism(tmp_path):
    """World size 4, two ranks empty in some steps, all of them in one, counts changing every step; both ends of every
    message go through dist.batch_isend_irecv (never a plain send / recv)."""
    port = _free_port()
    mp.spawn(_gather4_worker, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    assert all(open(tmp_path / f"ok{r}").read() == "1" for r in range(4))
