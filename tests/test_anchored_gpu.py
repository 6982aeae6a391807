"""GPU parity tests of the anchored stage (fused anchor-plane / composition / 6-of-8 window kernel,
host state machine, merges, dispatch order) against the CPU oracle and the fixtures.  Bit-exact."""
import glob
import os

import numpy as np
import pytest

import ribbit_amd
from cases import edge_cases, large_motif_cases, simulated_cases, structured_cases
from oracle_lib import LIST_ANCHORED, LIST_PERFECT, LIST_SUBST, Oracle

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL = edge_cases() + simulated_cases() + large_motif_cases() + structured_cases()


@pytest.mark.parametrize("name,seq,m_lo,m_hi", ALL, ids=[c[0] for c in ALL])
def test_anchored_stage_matches_oracle(name, seq, m_lo, m_hi):
    with ribbit_amd.Scanner(m_lo, m_hi) as sc, Oracle(seq, m_lo, m_hi) as o:
        sc.load_record(seq)
        o.run_all()
        got_calls = sc.anchored_calls()
        # composed planes XA_m (fasta_utils.cpp:143-161) as the kernel materialised them
        for m in sorted({m_lo, min(m_lo + 1, m_hi), (m_lo + m_hi) // 2, m_hi}):
            assert np.array_equal(sc.plane_bits(m), o.plane(m)), f"composed plane {m}"
        assert np.array_equal(got_calls.view("<i4"), o.calls(LIST_ANCHORED).view("<i4"))
        perfect, subst, anchored = sc.processShiftXORsAnchored()
        assert np.array_equal(perfect.view("<i4"), o.seeds(LIST_PERFECT).view("<i4"))
        assert np.array_equal(subst.view("<i4"), o.seeds(LIST_SUBST).view("<i4"))
        assert np.array_equal(anchored.view("<i4"), o.seeds(LIST_ANCHORED).view("<i4"))
        assert np.array_equal(sc.dispatch_seeds().view("<i4"), o.dispatch().view("<i4"))
        assert sc.guard_hits() == o.guard_hits()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))), ids=lambda p: os.path.basename(p)[:-4])
def test_anchored_stage_matches_fixture(path):
    g = np.load(path)
    with ribbit_amd.Scanner(int(g["m_lo"]), int(g["m_hi"])) as sc:
        sc.load_record(g["seq"].tobytes())
        assert np.array_equal(sc.anchored_calls().view("<i4"), g["anchored_calls"].view("<i4"))
        perfect, subst, anchored = sc.processShiftXORsAnchored()
        assert np.array_equal(perfect.view("<i4"), g["perfect"].view("<i4"))
        assert np.array_equal(subst.view("<i4"), g["subst"].view("<i4"))
        assert np.array_equal(anchored.view("<i4"), g["anchored"].view("<i4"))
        assert np.array_equal(sc.dispatch_seeds().view("<i4"), g["dispatch"].view("<i4"))


def test_all_composed_planes_on_a_dense_case():
    name, seq, m_lo, m_hi = [c for c in simulated_cases() if c[0] == "sim_m4_50_60k"][0]
    with ribbit_amd.Scanner(m_lo, m_hi) as sc, Oracle(seq, m_lo, m_hi) as o:
        sc.load_record(seq)
        o.run_perfect(); o.run_subst(); o.run_anchor_planes()
        sc.anchored_calls()
        for m in range(m_lo, m_hi + 1):
            assert np.array_equal(sc.plane_bits(m), o.plane(m)), f"composed plane {m}"


def test_anchored_stage_rejects_motifs_beyond_the_kernels_reach():
    with pytest.raises(ribbit_amd.RibbitHipError):
        ribbit_amd.Scanner(2, 1200)


@pytest.mark.parametrize("name,seq,m_lo,m_hi", large_motif_cases(), ids=[c[0] for c in large_motif_cases()])
def test_all_composed_planes_at_large_motifs(name, seq, m_lo, m_hi):
    with ribbit_amd.Scanner(m_lo, m_hi) as sc, Oracle(seq, m_lo, m_hi) as o:
        sc.load_record(seq)
        o.run_perfect(); o.run_subst(); o.run_anchor_planes()
        sc.anchored_calls()
        for m in range(m_lo, m_hi + 1, 7):
            assert np.array_equal(sc.plane_bits(m), o.plane(m)), f"composed plane {m}"


@pytest.mark.parametrize("name,seq,m_lo,m_hi", ALL, ids=[c[0] for c in ALL])
def test_seed_lists_without_asking_for_the_call_list_first(name, seq, m_lo, m_hi):
    """processShiftXORsAnchored straight after load: the anchored calls then only exist in their compact form
    (calls that fail the length filter are never materialised, only the largest end between two kept calls)."""
    with ribbit_amd.Scanner(m_lo, m_hi) as sc, Oracle(seq, m_lo, m_hi) as o:
        sc.load_record(seq)
        o.run_all()
        perfect, subst, anchored = sc.processShiftXORsAnchored()
        assert np.array_equal(perfect.view("<i4"), o.seeds(LIST_PERFECT).view("<i4"))
        assert np.array_equal(subst.view("<i4"), o.seeds(LIST_SUBST).view("<i4"))
        assert np.array_equal(anchored.view("<i4"), o.seeds(LIST_ANCHORED).view("<i4"))
        assert np.array_equal(sc.dispatch_seeds().view("<i4"), o.dispatch().view("<i4"))
        assert sc.guard_hits() == o.guard_hits()
        # and the full list is still available afterwards
        assert np.array_equal(sc.anchored_calls().view("<i4"), o.calls(LIST_ANCHORED).view("<i4"))


ANCHORED_SPAN = lambda m: int(0.9 * m) if m >= 10 else (m if m > 6 else 10)           # parse_anchored_shiftxor.cpp:572-573


@pytest.mark.parametrize("name,seq,m_lo,m_hi", ALL, ids=[c[0] for c in ALL])
def test_group_filter_of_the_scan_keeps_exactly_what_the_merge_needs(name, seq, m_lo, m_hi):
    """The compact form of the stage (what the product's merges consume): the scan kernel's group filter drops the groups of
    pass-streaks whose call cannot pass the length filter before they become events.  The kept calls, the end-of-sequence
    calls, the largest end of any call and the cursor bounds must be what the oracle's full call list implies -- and exactly
    what the same path gives with the filter switched off."""
    from ribbit_amd import STAGE_ANCHORED
    L = len(seq)
    with ribbit_amd.Scanner(m_lo, m_hi) as sc, Oracle(seq, m_lo, m_hi) as o:
        o.run_all()
        full = o.calls(LIST_ANCHORED)
        sc.load_record(seq)
        got = sc.stage_calls_chunk(STAGE_ANCHORED, 0, L + 1, 0, L)
        events_filtered = sc.last_event_count()
        os.environ["RIBBIT_NO_GROUP_FILTER"] = "1"
        try:
            sc.load_record(seq)
            plain = sc.stage_calls_chunk(STAGE_ANCHORED, 0, L + 1, 0, L)
            events_plain = sc.last_event_count()
        finally:
            del os.environ["RIBBIT_NO_GROUP_FILTER"]
    loop = full[full["pos"] < L]
    keep = (loop["end"] - loop["start"]) >= np.array([ANCHORED_SPAN(int(m)) for m in loop["mlen"]], dtype=np.int64) if len(loop) else np.zeros(0, bool)
    assert np.array_equal(got["calls"].view("<i4"), loop[keep].view("<i4"))
    assert np.array_equal(got["flush"].view("<i4"), full[full["pos"] >= L].view("<i4"))
    assert got["tail_pend"] == (int(loop["end"].max()) if len(loop) else -1)
    assert not got["inexact"] and events_filtered <= events_plain
    # cursor bounds: where one is given it is the largest end of any earlier call
    seen = np.concatenate(([-1], np.maximum.accumulate(loop["end"].astype(np.int64))[:-1]))[keep] if len(loop) else np.zeros(0, np.int64)
    for a in (got, plain):
        if a["pend"] is not None:
            given = a["pend"] >= 0
            assert np.array_equal(a["pend"][given], seen[given])
    assert np.array_equal(got["calls"].view("<i4"), plain["calls"].view("<i4")) and got["tail_pend"] == plain["tail_pend"]
    assert (got["pend"] is None) == (plain["pend"] is None) and (got["pend"] is None or np.array_equal(got["pend"], plain["pend"]))


def test_group_filter_cuts_the_event_volume():
    name, seq, m_lo, m_hi = [c for c in simulated_cases() if c[0] == "sim_cfg2_120k"][0]
    from ribbit_amd import STAGE_ANCHORED
    with ribbit_amd.Scanner(m_lo, m_hi) as sc:
        sc.load_record(seq)
        sc.anchored_calls()                         # the full call list: no filter
        unfiltered = sc.last_event_count()
        sc.load_record(seq)
        sc.stage_calls_chunk(STAGE_ANCHORED, 0, len(seq) + 1, 0, len(seq))
        filtered = sc.last_event_count()
    assert 0 < filtered < unfiltered / 3, (filtered, unfiltered)
