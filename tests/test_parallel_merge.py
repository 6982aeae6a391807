"""CPU tests of the range-parallel seed-list merges (ribbit_amd/csrc/parallel_merge.cpp): the window stages' calls are
cut into independent position ranges, merged on host threads and concatenated.  With the smallest range forced down to
a handful of calls (so that a record is cut wherever a cut is valid at all) the lists must still equal the oracle's,
which makes its calls strictly one after the other."""
import ctypes as C
import os

import numpy as np
import pytest

import ribbit_amd
from cases import edge_cases, large_motif_cases, simulated_cases, structured_cases
from fuzz import fuzz_case
from oracle_lib import LIST_ANCHORED, LIST_PERFECT, LIST_SUBST, Oracle

CASES = edge_cases() + simulated_cases() + large_motif_cases() + [c for c in structured_cases() if "homopolymer" not in c[0]]


@pytest.fixture(params=[1, 5, 64], ids=lambda v: f"min_range_{v}")
def tiny_ranges(request):
    lib = ribbit_amd.load_library()
    old = os.environ.get("RIBBIT_THREADS")
    os.environ["RIBBIT_THREADS"] = "4"
    lib.ribbit_debug_set_merge_min_range(request.param)
    yield request.param
    lib.ribbit_debug_set_merge_min_range(4096)
    if old is None:
        del os.environ["RIBBIT_THREADS"]
    else:
        os.environ["RIBBIT_THREADS"] = old


def _check(seq, m_lo, m_hi, tag):
    with Oracle(seq, m_lo, m_hi) as o:
        o.run_all()
        r = ribbit_amd.host_replay_calls(m_lo, m_hi, seq, o.calls(LIST_PERFECT), o.calls(LIST_SUBST), o.calls(LIST_ANCHORED))
        assert np.array_equal(r["perfect"].view("<i4"), o.seeds(LIST_PERFECT).view("<i4")), tag
        assert np.array_equal(r["subst"].view("<i4"), o.seeds(LIST_SUBST).view("<i4")), tag
        assert np.array_equal(r["anchored"].view("<i4"), o.seeds(LIST_ANCHORED).view("<i4")), tag
        assert np.array_equal(r["dispatch"].view("<i4"), o.dispatch().view("<i4")), tag
        assert r["guard_hits"] == o.guard_hits(), tag
    lib = ribbit_amd.load_library()
    merged = (C.c_int32 * 5)()
    lib.ribbit_debug_last_merge(1, C.byref(merged))
    return int(merged[0]), int(lib.ribbit_debug_last_dispatch_ranges())       # ranges of the anchored merge, of the dispatch merge


@pytest.mark.parametrize("name,seq,m_lo,m_hi", CASES, ids=[c[0] for c in CASES])
def test_range_parallel_merges_equal_the_oracles_lists(tiny_ranges, name, seq, m_lo, m_hi):
    _check(seq, m_lo, m_hi, name)


def test_the_dispatch_merge_runs_over_the_same_cuts(tiny_ranges):
    """the 3-way dispatch merge splits at the anchored stage's cuts (checked at run time: every list must split cleanly
    there, else it runs sequentially): on the simulated records it does use them, and the order equals the oracle's"""
    used = []
    for name, seq, m_lo, m_hi in simulated_cases():
        used.append(_check(seq, m_lo, m_hi, name))
    assert all(d == m or d == 1 for m, d in used), used          # either all of the stage's ranges or the sequential merge
    assert sum(1 for m, d in used if m > 1 and d == m) >= len(used) // 2, used


@pytest.mark.parametrize("block", range(6))
def test_range_parallel_merges_on_fuzzed_records(tiny_ranges, block):
    for seed in range(9000 + 40 * block, 9000 + 40 * (block + 1)):
        seq, m_lo, m_hi = fuzz_case(seed)
        _check(seq, m_lo, m_hi, f"seed {seed}: {len(seq)} bases, -m {m_lo} -M {m_hi}")


@pytest.mark.parametrize("calls_per_range", [1, 4, 64])
def test_the_ranges_of_the_device_pass_merged_on_the_host_threads(calls_per_range):
    """The GPU's pass of the anchored merge (anchored_merge.hip) cuts the stage into ranges of a few dozen calls -- prepared on the
    host threads: cuts searched in pieces, the ranges' cursors by bisection on the running maximum of the list's starts -- gives
    the long ones to the host threads while its kernel runs and the ones it cannot merge afterwards.  RIBBIT_MERGE_DEVICE_RANGES
    runs all of that with a device that fails after the host's share: the same lists as the oracle's."""
    lib = ribbit_amd.load_library()
    old = {k: os.environ.get(k) for k in ("RIBBIT_MERGE_DEVICE_RANGES", "RIBBIT_THREADS")}
    os.environ.update(RIBBIT_MERGE_DEVICE_RANGES=str(calls_per_range), RIBBIT_THREADS="4")
    lib.ribbit_debug_set_merge_min_range(calls_per_range)
    try:
        most = 0
        for name, seq, m_lo, m_hi in CASES:
            most = max(most, _check(seq, m_lo, m_hi, name)[0])
        for seed in range(9000, 9060):
            seq, m_lo, m_hi = fuzz_case(seed)
            most = max(most, _check(seq, m_lo, m_hi, f"seed {seed}")[0])
        for seed in (20037, 20171):          # (a range that reads a type its left neighbour changes afterwards: the validation walk)
            seq, m_lo, m_hi = fuzz_case(seed)
            _check(seq, m_lo, m_hi, f"seed {seed}")
        from ribbit_amd.simulate import simulate_sequence
        _check(simulate_sequence(300_000, 38, 2, 30)[0], 2, 30, "seed 38 (a list-head write that changes its entry)")
        assert most > 100, most
    finally:
        lib.ribbit_debug_set_merge_min_range(4096)
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_a_two_megabase_record_is_cut_into_many_ranges():
    from ribbit_amd.simulate import simulate_sequence
    seq, _ = simulate_sequence(400_000, 17, 2, 40, n_block_rate=0.3)
    os.environ["RIBBIT_THREADS"] = "8"
    try:
        _check(seq, 2, 40, "400 kb, default range size")
    finally:
        del os.environ["RIBBIT_THREADS"]


def test_redo_in_order_restores_what_the_ranges_changed(tiny_ranges):
    """The fallback (a list-head write that changes an entry, or an empty first range): the parallel pass has retired
    seeds of the earlier stages by then; they must be live again when the stage is redone in order."""
    os.environ["RIBBIT_MERGE_FORCE_REDO"] = "1"
    try:
        for name, seq, m_lo, m_hi in [c for c in CASES if c[0] in ("sim_m4_50_60k", "n_runs", "M200_mixed")]:
            _check(seq, m_lo, m_hi, name)
        out = (C.c_int32 * 5)()
        ribbit_amd.load_library().ribbit_debug_last_merge(1, C.byref(out))
        assert out[2] & 1 == 1
    finally:
        del os.environ["RIBBIT_MERGE_FORCE_REDO"]


def test_ranges_that_read_a_stale_type_are_merged_again(tiny_ranges):
    """Seeds 20037 / 20171 (found by scanning): a range's candidate walk meets a seed of the range before that is retired
    later in that range; run in parallel it may see it live, which the validation pass must catch."""
    for seed in (20037, 20171):
        seq, m_lo, m_hi = fuzz_case(seed)
        _check(seq, m_lo, m_hi, f"seed {seed}")


def test_a_list_head_write_that_only_mattered_inside_its_own_range(tiny_ranges):
    """Fuzz seed 430991 (2 kb, -m 7 -M 28; found late in round 4 by a sweep with ranges of one call): a call of the FIRST range
    writes a list-head entry (Q8, parse_anchored_shiftxor.cpp:511-522) that is live at that moment and that an ordinary merge of the
    same range retires a few calls later.  The reference makes the write at once, so the calls in between see the entry retired; a
    worker only logs it, and by the time the ranges are walked the logged write looks like a no-op.  The log now says whether the
    write changed its target when it was made (ListRefs::HeadWrite::changed_then) and such a range is merged again with its writes
    made.  Both the host threads' ranges and the ranges of the device pass (with a device that cannot run)."""
    seq, m_lo, m_hi = fuzz_case(430991)
    _check(seq, m_lo, m_hi, "seed 430991")
    os.environ["RIBBIT_MERGE_DEVICE_RANGES"] = str(tiny_ranges)
    try:
        _check(seq, m_lo, m_hi, "seed 430991, device ranges")
    finally:
        del os.environ["RIBBIT_MERGE_DEVICE_RANGES"]


def test_a_list_head_write_that_changes_its_entry_keeps_the_merge_parallel_and_exact(tiny_ranges):
    """Q8 (parse_anchored_shiftxor.cpp:511-522): the coverage code writes an entry at the HEAD of the perfect / substitution
    list from anywhere in the record.  On this record (generator seed 38, 300 kb, -M 30: found by a sweep, such writes are rare
    below chromosome size) one write changes its entry.  Until round 3 that sent the whole stage back to one thread; now the
    range that makes it is done again with the write made, the lists still equal the oracle's, and the stage is not redone."""
    from ribbit_amd.simulate import simulate_sequence
    seq, _ = simulate_sequence(300_000, 38, 2, 30)
    _check(seq, 2, 30, "seed 38")
    out = (C.c_int32 * 5)()
    ribbit_amd.load_library().ribbit_debug_last_merge(1, C.byref(out))
    ranges, again, in_order, changing, tail = (int(x) for x in out)
    assert ranges > 1 and changing >= 1 and again >= 1 and in_order & 1 == 0 and (tail >> 8) >= 1, list(out)
    # Behind the change only the ranges that can come out differently run again (those that read a changed entry by loop
    # counter, or whose left cut lies within a motif of where the entry was or went): with every range behind it run again,
    # as until round 3 (the variable is a test hook), the lists are the same and there are more range runs.
    passes, runs = tail >> 8, in_order >> 1
    os.environ["RIBBIT_MERGE_RERUN_ALL"] = "1"
    try:
        _check(seq, 2, 30, "seed 38, every range behind the change again")
        ribbit_amd.load_library().ribbit_debug_last_merge(1, C.byref(out))
    finally:
        del os.environ["RIBBIT_MERGE_RERUN_ALL"]
    assert (int(out[4]) >> 8) == passes and int(out[2]) >> 1 >= runs, (list(out), passes, runs)
    if passes > 1 and ranges > 8:
        assert int(out[2]) >> 1 > runs, (list(out), runs)
