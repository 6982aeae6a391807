"""CPU tests of the chunk-sharded host pipeline: events (numpy restatement of the kernels' output, built from
the oracle's planes) are split into parts by position, optionally exchanged between two gloo ranks, and
replayed by ribbit_host_scan_from_events; the lists must equal the oracle's."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import pyevents
import ribbit_amd
from cases import edge_cases, simulated_cases
from oracle_lib import LIST_ANCHORED, LIST_PERFECT, LIST_SUBST, Oracle
from ribbit_amd import sharded

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [c for c in edge_cases() if len(c[1]) >= 64] + simulated_cases()[:3]


def _oracle_parts(seq, m_lo, m_hi, nparts):
    """(parts as ranks would produce them, oracle lists)"""
    with Oracle(seq, m_lo, m_hi) as o:
        ev0 = pyevents.perfect_events(o, m_lo, m_hi)
        ev1 = pyevents.window_events(o, m_lo, m_hi, 1)
        o.run_perfect(); o.run_subst(); o.run_anchor_planes()
        ev2 = pyevents.window_events(o, m_lo, m_hi, 2)
        xa_full, stride = ribbit_amd.pack_bit_planes([o.plane(m) for m in range(m_lo, m_hi + 1)], len(seq))
        o.run_anchored(); o.run_dispatch()
        want = {"perfect": o.seeds(LIST_PERFECT), "subst": o.seeds(LIST_SUBST), "anchored": o.seeds(LIST_ANCHORED),
                "dispatch": o.dispatch(), "guard_hits": o.guard_hits()}
    hi, lo, brk = ribbit_amd.pack_planes(seq, m_hi)
    parts = []
    for own_lo, own_hi, _, _ in sharded.plan_chunks(len(seq), nparts, m_hi):
        w0, w1 = own_lo // 32, min((own_hi + 31) // 32, len(seq) // 32 + 1)
        p = {"own_lo": own_lo, "own_hi": own_hi, "hi": hi[w0:w1].copy(), "lo": lo[w0:w1].copy(), "brk": brk[w0:w1].copy(),
             "xa": xa_full[:, w0:w1].copy()}
        for k, (ev, cnt) in enumerate((ev0, ev1, ev2)):
            p[f"ev{k}"], p[f"cnt{k}"] = pyevents.split_events(ev, cnt, own_lo, own_hi)
        parts.append(p)
    return parts, want


def _same(got, want):
    for k in ("perfect", "subst", "anchored", "dispatch"):
        assert np.array_equal(got[k].view("<i4"), want[k].view("<i4")), k
    assert got["guard_hits"] == want["guard_hits"]


@pytest.mark.parametrize("name,seq,m_lo,m_hi", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("nparts", [1, 3])
def test_pipeline_from_events_matches_oracle(name, seq, m_lo, m_hi, nparts):
    parts, want = _oracle_parts(seq, m_lo, m_hi, nparts)
    _same(sharded.merge_parts(m_lo, m_hi, len(seq), parts), want)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    name, seq, m_lo, m_hi = simulated_cases()[2]
    parts, want = _oracle_parts(seq, m_lo, m_hi, world)      # every rank could scan only its own chunk; here they are precomputed
    gathered = sharded.gather_parts(parts[rank])            # only rank 0 receives (and merges)
    assert (gathered is None) == (rank != 0)
    if rank == 0:
        got = sharded.merge_parts(m_lo, m_hi, len(seq), gathered)
        _same(got, want)
        np.save(os.path.join(out_dir, "dispatch0.npy"), got["dispatch"])
    dist.barrier()
    dist.destroy_process_group()


def test_chunk_sharded_exchange_world2(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert len(np.load(tmp_path / "dispatch0.npy")) > 0


def test_pair_halves_pairs_by_motif_and_position_and_rejects_garbage():
    ev = lambda pos, m, kind: np.uint64(pos | (m << 32) | (kind << 48))
    halves = np.array([ev(500, 3, 1), ev(100, 3, 0), ev(90, 2, 0), ev(7000, 2, 3)], dtype=np.uint64)
    runs = ribbit_amd.pair_halves(halves)
    assert [tuple(int(x) for x in r) for r in runs] == [(90, 7000, 2, 2), (100, 500, 3, 0)]
    assert len(ribbit_amd.pair_halves(np.zeros(0, np.uint64))) == 0
    with pytest.raises(ribbit_amd.RibbitHipError):
        ribbit_amd.pair_halves(np.array([ev(1, 2, 0), ev(5, 2, 0)], dtype=np.uint64))
