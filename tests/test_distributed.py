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
