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
