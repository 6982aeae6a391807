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
