"""GPU tests of chunk-sharding: one GPU plays every rank in turn (loads chunk + halos, runs the device side of all three
stages on the piece, keeps the calls whose scan position it owns and its own plane words); the merge of what the chunks
kept must equal the oracle's lists -- and the BED its pipeline's -- for any number of parts."""
import numpy as np
import pytest

import ribbit_amd
from cases import edge_cases, simulated_cases
from oracle_lib import LIST_ANCHORED, LIST_PERFECT, LIST_SUBST, Oracle
from ribbit_amd import STAGE_ANCHORED, STAGE_SUBST, sharded

pytestmark = pytest.mark.gpu
CASES = [c for c in edge_cases() if len(c[1]) >= 64] + simulated_cases()


@pytest.mark.parametrize("name,seq,m_lo,m_hi", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("nparts", [2, 4, 7])
def test_chunk_sharded_scan_matches_oracle(name, seq, m_lo, m_hi, nparts):
    with Oracle(seq, m_lo, m_hi) as o:
        o.run_all()
        want = {"perfect": o.seeds(LIST_PERFECT), "subst": o.seeds(LIST_SUBST), "anchored": o.seeds(LIST_ANCHORED),
                "dispatch": o.dispatch()}
        want_bed = o.refine_bed(name)
    with ribbit_amd.Scanner(m_lo, m_hi) as sc:
        parts = [sharded.scan_part(sc, seq, plan) for plan in sharded.plan_chunks(len(seq), nparts, m_hi)]
    got = sharded.merge_parts(m_lo, m_hi, len(seq), parts)
    for k in want:
        assert np.array_equal(got[k].view("<i4"), want[k].view("<i4")), k
    # nothing but 16 bytes per kept call (and 4 per call of a chunk that holds a call made at an N) leaves a chunk for
    # the window stages: no event, no streak record
    for p in parts:
        b = sharded.part_bytes(p)
        assert b["window_call_bytes"] == 16 * b["kept_window_calls"]
        assert b["records"] <= 20 * b["kept_window_calls"] + 16 * (b["perfect_runs"] + len(p["halves"]) + len(p["subst_flush"]) + len(p["anchored_flush"]))
    hi, lo, brk, xa, stride = got["planes"]
    assert ribbit_amd.host_refine_bed(m_lo, m_hi, seq, xa, stride, got["dispatch"], name).split("\n") == want_bed.split("\n")


def test_every_call_is_kept_by_exactly_one_chunk_and_a_short_halo_is_noticed():
    """the chunks' kept calls back to back ARE the record's kept calls (same order, same cursor bounds where they matter);
    a piece whose left halo is shorter than the repeat that straddles its cut says so instead of returning a cut group"""
    name, seq, m_lo, m_hi = [c for c in edge_cases() if c[0] == "long_run_cross_tiles"][0]
    seq = seq + simulated_cases()[1][1][:80_000]
    L = len(seq)
    with ribbit_amd.Scanner(m_lo, m_hi) as sc:
        sc.load_record(seq)
        whole = {st: sc.stage_calls_chunk(st, 0, L + 1, 0, L) for st in (STAGE_SUBST, STAGE_ANCHORED)}
        assert not any(w["inexact"] for w in whole.values())
        plans = sharded.plan_chunks(L, 5, m_hi, left_halo=0)          # the smallest halo the kernels' reach allows
        # chunk 1 starts inside the 40-kb repeat: with the minimal halo its first group is cut, and it must say so
        own_lo, own_hi, load_lo, load_hi = plans[1]
        sc.load_record(seq[load_lo:load_hi])
        cut = sc.stage_calls_chunk(STAGE_ANCHORED, own_lo - load_lo, own_hi - load_lo, load_lo, L)
        assert cut["inexact"]
        parts = [sharded.scan_part(sc, seq, plan) for plan in plans]
    assert max(p["halo_grown"] for p in parts) >= 1 and parts[0]["halo_grown"] == 0
    for st, key in ((STAGE_SUBST, "subst"), (STAGE_ANCHORED, "anchored")):
        calls = np.concatenate([p[f"{key}_calls"] for p in parts])
        assert np.array_equal(calls.view("<i4"), whole[st]["calls"].view("<i4")), key
        flush = np.concatenate([p[f"{key}_flush"] for p in parts])
        assert np.array_equal(flush.view("<i4"), whole[st]["flush"].view("<i4")), key
        assert max(p[f"{key}_tail_pend"] for p in parts) == whole[st]["tail_pend"]


def test_chunk_geometry_is_checked():
    name, seq, m_lo, m_hi = simulated_cases()[1]
    L = len(seq)
    with ribbit_amd.Scanner(m_lo, m_hi) as sc:
        sc.load_record(seq[1000:60_000])
        with pytest.raises(ribbit_amd.RibbitHipError, match="beyond the chunk"):
            sc.stage_calls_chunk(STAGE_SUBST, 2000, 58_900, 1000, L)                  # right halo too short
        with pytest.raises(ribbit_amd.RibbitHipError, match="before the chunk"):
            sc.stage_calls_chunk(STAGE_SUBST, 10, 30_000, 1000, L)                     # left halo shorter than the kernels' reach
        with pytest.raises(ribbit_amd.RibbitHipError, match="geometry"):
            sc.stage_calls_chunk(STAGE_SUBST, 2000, 30_000, L, L)                      # piece beyond the record


@pytest.mark.parametrize("nparts", [4, 7])      # cuts at 30 kb resp. 17/34 kb: inside the 40-kb repeat
def test_chunk_local_pairing_plus_edge_halves_gives_the_whole_records_runs(nparts):
    name, seq, m_lo, m_hi = [c for c in edge_cases() if c[0] == "long_run_cross_tiles"][0]
    seq = seq + simulated_cases()[1][1][:80_000]
    with ribbit_amd.Scanner(m_lo, m_hi) as sc:
        sc.load_record(seq)
        want = sc.scan_perfect_runs()
        runs, halves = [], []
        for own_lo, own_hi, load_lo, load_hi in sharded.plan_chunks(len(seq), nparts, m_hi):
            sc.load_record(seq[load_lo:load_hi])
            r, h = sc.perfect_runs_partial(own_lo - load_lo, own_hi - load_lo, load_lo)
            runs.append(r); halves.append(h)
    # the 40-kb perfect repeat crosses chunk edges: its run must come out of the edge pairing
    assert sum(len(h) for h in halves) > 0
    got = np.sort(np.concatenate(runs + [ribbit_amd.pair_halves(np.concatenate(halves))]), order=["mlen", "start"])
    assert np.array_equal(got.view("<i4"), want.view("<i4"))


@pytest.mark.parametrize("nparts", [1, 4, 7])
def test_device_paired_chunk_records_plus_halves_give_the_whole_records_runs(nparts):
    """ribbit_hip_scan_perfect_chunk (pairing on the GPU, own-range classification in the kernel) against the whole
    record's scan and against the host-side chunk pairing of ribbit_hip_perfect_runs_partial."""
    name, seq, m_lo, m_hi = [c for c in edge_cases() if c[0] == "long_run_cross_tiles"][0]
    seq = seq + simulated_cases()[1][1][:80_000]
    with ribbit_amd.Scanner(m_lo, m_hi) as sc:
        sc.load_record(seq)
        want = sc.scan_perfect_runs()
        parts, halves, n_half_old = [], [], 0
        for own_lo, own_hi, load_lo, load_hi in sharded.plan_chunks(len(seq), nparts, m_hi):
            sc.load_record(seq[load_lo:load_hi])
            rec, hv = sc.scan_perfect_chunk(own_lo - load_lo, own_hi - load_lo, load_lo)
            rec = rec.copy()
            old_runs, old_halves = sc.perfect_runs_partial(own_lo - load_lo, own_hi - load_lo, load_lo)
            assert np.array_equal(rec[rec["term"] >= 0].view("<i4"), old_runs.view("<i4"))
            assert set(np.unique(rec["term"])) <= {-1, 0, 1, 2} and len(hv) == len(old_halves)
            parts.append(rec); halves.append(hv); n_half_old += len(old_halves)
    assert (n_half_old > 0) == (nparts > 1)
    got = ribbit_amd.merge_chunk_runs(parts, halves)
    assert np.array_equal(got.view("<i4"), want.view("<i4"))


def test_chunk_records_land_in_a_page_locked_node_shared_segment():
    from ribbit_amd.node_gather import NodeGather
    name, seq, m_lo, m_hi = simulated_cases()[1]
    with ribbit_amd.Scanner(m_lo, m_hi) as sc:
        sc.load_record(seq)
        want = sc.scan_perfect_runs()
        ng = NodeGather(ribbit_amd.RUN_DT, len(want) + 8, 2 * (m_hi - m_lo + 1), rank=0, world=1)
        try:
            for addr, nbytes in ng.my_cells():
                sc.host_register(addr, nbytes)
            for k in (1, 2, 3):
                if k > 1:
                    ng.release(k - 1)
                ng.wait_free(k)
                rec, hv = ng.mine(k)
                n, nh = sc.scan_perfect_chunk(0, len(seq) + 1, 0, out=rec, halves_out=hv)
                ng.publish(k, n, nh)
                parts, halves = ng.collect(k)
                assert nh == 0 and np.array_equal(parts[0].view("<i4"), want.view("<i4"))
            small = np.zeros(3, ribbit_amd.RUN_DT)
            with pytest.raises(ribbit_amd.RibbitHipError, match="do not fit"):
                sc.scan_perfect_chunk(0, len(seq) + 1, 0, out=small, halves_out=hv)
            for addr, _ in ng.my_cells():
                sc.host_unregister(addr)
        finally:
            parts = halves = rec = hv = None
            ng.close()


def test_run_records_can_be_taken_from_device_memory_without_a_host_copy():
    """ribbit_hip_scan_perfect_end_device: the run records where the pairing kernels left them, viewed as a torch tensor
    (what the RCCL gather of the N > 1 bench sends GPU-to-GPU) equal the ones the ordinary end copies to the host."""
    import torch
    from ribbit_amd.distributed import device_bytes
    name, seq, m_lo, m_hi = [c for c in simulated_cases() if c[0] == "sim_cfg2_120k"][0]
    dev = torch.device("cuda", 0)
    with ribbit_amd.Scanner(m_lo, m_hi) as sc:
        sc.load_record(seq)
        want = sc.scan_perfect_runs()
        sc.load_record(seq)
        sc.scan_perfect_begin()
        ptr, n, hptr, nh = sc.scan_perfect_end_device()
        assert n == len(want) and nh == 0
        got = device_bytes(ptr, n * 16, dev).cpu().numpy().view(ribbit_amd.RUN_DT)
        assert np.array_equal(got.view("<i4"), want.view("<i4"))
        # the handle is usable afterwards
        assert np.array_equal(sc.scan_perfect_runs().view("<i4"), want.view("<i4"))
