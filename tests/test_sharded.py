"""CPU tests of the merging rank's half of the chunk-sharded path (ribbit_host_merge_chunks, ribbit_amd.sharded): the
oracle's call lists are cut into what the chunks of a record would keep -- kept calls by owned scan position with
chunk-local cursor bounds, perfect runs with the halves a chunk edge cuts, plane words -- optionally exchanged between
gloo ranks, and merged; the lists must equal the oracle's."""
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
from ribbit_amd import CALL_DT, RUN_DT, sharded

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [c for c in edge_cases() if len(c[1]) >= 64] + simulated_cases()[:3]
SUBST_SPAN = lambda m: m // 3 if m > 30 else 10                                       # parse_substitute_shiftxor.cpp:423
ANCHORED_SPAN = lambda m: int(0.9 * m) if m >= 10 else (m if m > 6 else 10)           # parse_anchored_shiftxor.cpp:572-573


def _chunk_calls(calls, length, own_lo, own_hi, last, span):
    """what ribbit_hip_stage_calls_chunk keeps of a stage's full call list for the chunk owning scan positions
    [own_lo, own_hi): the calls that pass the length filter, each with the largest end of the chunk's earlier calls;
    the largest end of any of its calls; the end-of-sequence calls (pos == length) if it is the last chunk"""
    loop = calls[(calls["pos"] < length) & (calls["pos"] >= own_lo) & (calls["pos"] < own_hi)]
    ends = loop["end"].astype(np.int64)
    seen = np.concatenate(([-1], np.maximum.accumulate(ends)[:-1])) if len(loop) else np.zeros(0, np.int64)
    keep = (loop["end"] - loop["start"]) >= np.array([span(int(m)) for m in loop["mlen"]], dtype=np.int64) if len(loop) else np.zeros(0, bool)
    flush = calls[calls["pos"] >= length] if last else np.zeros(0, CALL_DT)
    return loop[keep].copy(), seen[keep].astype("<i4"), int(ends.max()) if len(loop) else -1, flush.copy()


def _chunk_runs(runs, own_lo, own_hi):
    """ribbit_hip_scan_perfect_chunk's records for the chunk: complete runs it owns, halves of the runs its edges cut"""
    s, e = runs["start"], runs["end"]
    s_own, e_own = (s >= own_lo) & (s < own_hi), (e >= own_lo) & (e < own_hi)
    whole = runs[s_own & (e < own_hi)].copy()
    hs = runs[s_own & (e >= own_hi)].copy(); hs["end"] = -1; hs["term"] = ribbit_amd.RUN_HALF_START
    he = runs[e_own & (s < own_lo)].copy(); he["start"] = -1; he["term"] = ribbit_amd.RUN_HALF_END + he["term"]
    return whole, np.concatenate((hs, he))


def _oracle_parts(seq, m_lo, m_hi, nparts):
    """(parts as ranks would produce them, oracle lists)"""
    L = len(seq)
    with Oracle(seq, m_lo, m_hi) as o:
        ev0, cnt0 = pyevents.perfect_events(o, m_lo, m_hi)
        runs = ribbit_amd.host_perfect_runs_from_events(m_lo, m_hi, [ev0], [cnt0])
        o.run_perfect(); o.run_subst(); o.run_anchor_planes()
        xa_full, stride = ribbit_amd.pack_bit_planes([o.plane(m) for m in range(m_lo, m_hi + 1)], L)
        o.run_anchored(); o.run_dispatch()
        subst, anchored = o.calls(LIST_SUBST), o.calls(LIST_ANCHORED)
        want = {"perfect": o.seeds(LIST_PERFECT), "subst": o.seeds(LIST_SUBST), "anchored": o.seeds(LIST_ANCHORED),
                "dispatch": o.dispatch(), "guard_hits": o.guard_hits()}
    hi, lo, brk = ribbit_amd.pack_planes(seq, m_hi)
    parts = []
    plans = sharded.plan_chunks(L, nparts, m_hi)
    for k, (own_lo, own_hi, _, _) in enumerate(plans):
        w0, w1 = own_lo // 32, min((own_hi + 31) // 32, L // 32 + 1)
        p = {"own_lo": own_lo, "own_hi": own_hi, "halo_grown": 0, "left_halo": 0,
             "hi": hi[w0:w1].copy(), "lo": lo[w0:w1].copy(), "brk": brk[w0:w1].copy(), "xa": xa_full[:, w0:w1].copy()}
        p["runs"], p["halves"] = _chunk_runs(runs, own_lo, own_hi)
        for name, calls, span in (("subst", subst, SUBST_SPAN), ("anchored", anchored, ANCHORED_SPAN)):
            c, pend, tail, flush = _chunk_calls(calls, L, own_lo, own_hi, k == nparts - 1, span)
            p[f"{name}_calls"], p[f"{name}_pend"], p[f"{name}_tail_pend"], p[f"{name}_flush"] = c, pend, tail, flush
        parts.append(p)
    return parts, want


def _same(got, want):
    for k in ("perfect", "subst", "anchored", "dispatch"):
        assert np.array_equal(got[k].view("<i4"), want[k].view("<i4")), k
    assert got["guard_hits"] == want["guard_hits"]


@pytest.mark.parametrize("name,seq,m_lo,m_hi", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("nparts", [1, 3, 8])
def test_merge_of_the_chunks_kept_calls_matches_oracle(name, seq, m_lo, m_hi, nparts):
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
    # the same parts through the node-shared segment: the same lists (or, everywhere alike, "no segment here")
    ok, shared = sharded.gather_parts_shm(parts[rank])
    assert (shared is None) == (rank != 0 or not ok)
    if rank == 0 and ok:
        for a, b in zip(gathered, shared):
            assert a.keys() == b.keys()
            for k in a:
                if a[k] is None or b[k] is None or np.ndim(a[k]) == 0:
                    assert a[k] == b[k], k
                else:
                    assert np.asarray(a[k]).tobytes() == np.asarray(b[k]).tobytes(), k
        _same(sharded.merge_parts(m_lo, m_hi, len(seq), shared), want)
        np.save(os.path.join(out_dir, "shm_ok.npy"), np.ones(1))
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
