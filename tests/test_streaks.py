"""CPU check of the rule the GPU window stage implements (ribbit_amd/csrc/window_stage.hip, restated in numpy in
tests/pystreaks.py): the reference's sequential per-motif window state machine, as the oracle walks it base by base,
equals "one call per group of joined pass-streaks" -- same calls, same call order -- and the cursor bounds of the
compact form move the merges' cursors exactly as the full call list would."""
import numpy as np
import pytest

from cases import edge_cases, simulated_cases, structured_cases
from fuzz import fuzz_case
from oracle_lib import LIST_ANCHORED, LIST_SUBST, Oracle
from pyevents import window_events
from pystreaks import calls_from_streaks, compact_bounds, streaks_of

SMALL = edge_cases() + [c for c in simulated_cases() if c[0] in ("sim_m4_50_60k",)] + \
    [c for c in structured_cases() if c[0] in ("mostly_n", "dinucleotide_60k_sparse_mismatches")]


def subst_cutoff(m):        # parse_substitute_shiftxor.cpp:423
    return m // 3 if m > 30 else 10


def anchored_cutoff(m):     # parse_anchored_shiftxor.cpp:572-573
    return int(0.9 * m) if m >= 10 else (m if m > 6 else 10)


def check_stage(o, m_lo, m_hi, allowed, which, cutoff):
    ev, cnt = window_events(o, m_lo, m_hi, allowed)
    nmask = o.nmask()
    inloop, flush, edge = calls_from_streaks(streaks_of(ev, cnt, m_lo), nmask)
    want = o.calls(which)
    got = np.array(inloop + flush, dtype=np.int32).reshape(-1, 4)
    assert np.array_equal(got, want.view("<i4").reshape(-1, 4))
    # compact form: what the cursors see before every kept call
    kept, bounds = compact_bounds(inloop, edge, cutoff)
    k = 0
    seen_all = -1          # largest end of any earlier call (what the full replay has advanced the cursors by)
    seen_compact = -1      # the same from the kept calls and their bounds alone
    for c in inloop:
        _, m, s, e = c
        if e - s >= cutoff(m):
            assert kept[k] == c
            seen_compact = max(seen_compact, bounds[k])
            assert max(seen_all, e) == max(seen_compact, e), f"cursor bound of kept call {c}"
            seen_compact = max(seen_compact, e)
            k += 1
        seen_all = max(seen_all, e)
    assert k == len(kept)
    return len(inloop), sum(edge)


@pytest.mark.parametrize("name,seq,m_lo,m_hi", SMALL, ids=[c[0] for c in SMALL])
def test_streak_rule_equals_the_sequential_state_machine(name, seq, m_lo, m_hi):
    with Oracle(seq, m_lo, m_hi) as o:
        o.run_perfect()
        o.run_subst()
        check_stage(o, m_lo, m_hi, 1, LIST_SUBST, subst_cutoff)      # planes are still the plain X_m here
        o.run_anchor_planes()
        o.run_anchored()
        check_stage(o, m_lo, m_hi, 2, LIST_ANCHORED, anchored_cutoff)


@pytest.mark.parametrize("seed", range(60))
def test_streak_rule_on_fuzzed_records(seed):
    seq, m_lo, m_hi = fuzz_case(seed)
    if len(seq) > 6000:
        seq = seq[:6000]
    with Oracle(seq, m_lo, m_hi) as o:
        o.run_perfect()
        o.run_subst()
        o.run_anchor_planes()
        o.run_anchored()
        check_stage(o, m_lo, m_hi, 2, LIST_ANCHORED, anchored_cutoff)


def test_n_dense_record_has_edge_calls():
    """N every few dozen bases: most calls are made next to an N, where the bound of a call is not implied by
    its own end"""
    rs = np.random.RandomState(5)
    seq = bytearray((b"CAGCAGCAGCATCAGCAG" * 400) + bytes(rs.choice(list(b"ACGT"), 3000)))
    for p in rs.randint(0, len(seq), 300):
        seq[p] = ord("N")
    with Oracle(bytes(seq), 2, 12) as o:
        o.run_perfect(); o.run_subst(); o.run_anchor_planes(); o.run_anchored()
        n, n_edge = check_stage(o, 2, 12, 2, LIST_ANCHORED, anchored_cutoff)
        assert n_edge > 50
