"""GPU parity tests of the anchored stage's merge as device work (anchored_merge.hip: one lane per independent range of the
stage's kept calls; parse_anchored_shiftxor.cpp:113-534, merge_types.cpp:11-189): with the thresholds lowered so that every
record takes the device pass, the three seed lists, the dispatch order and the guard count equal the CPU oracle's, for ranges
of 1, 8 and 64 calls.  Bit-exact."""
import os

import numpy as np
import pytest

import ribbit_amd
from cases import edge_cases, large_motif_cases, simulated_cases, structured_cases
from fuzz import fuzz_case
from oracle_lib import LIST_ANCHORED, LIST_PERFECT, LIST_SUBST, Oracle

pytestmark = pytest.mark.gpu
ALL = edge_cases() + simulated_cases() + large_motif_cases() + structured_cases()


@pytest.fixture(params=[1, 8, 64])
def device_ranges(request):
    lib = ribbit_amd.load_library()
    old = {k: os.environ.get(k) for k in ("RIBBIT_DEVICE_MERGE_MIN", "RIBBIT_DEVICE_MERGE_RANGE", "RIBBIT_THREADS")}
    os.environ.update(RIBBIT_DEVICE_MERGE_MIN="1", RIBBIT_DEVICE_MERGE_RANGE=str(request.param), RIBBIT_THREADS="4")
    lib.ribbit_debug_set_merge_min_range(request.param)
    yield request.param
    lib.ribbit_debug_set_merge_min_range(4096)
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _check(seq, m_lo, m_hi, tag):
    with ribbit_amd.Scanner(m_lo, m_hi) as sc, Oracle(seq, m_lo, m_hi) as o:
        sc.load_record(seq)
        o.run_all()
        perfect, subst, anchored = sc.processShiftXORsAnchored()
        stats = ribbit_amd.last_device_merge()
        assert np.array_equal(perfect.view("<i4"), o.seeds(LIST_PERFECT).view("<i4")), tag
        assert np.array_equal(subst.view("<i4"), o.seeds(LIST_SUBST).view("<i4")), tag
        assert np.array_equal(anchored.view("<i4"), o.seeds(LIST_ANCHORED).view("<i4")), tag
        assert np.array_equal(sc.dispatch_seeds().view("<i4"), o.dispatch().view("<i4")), tag
        assert sc.guard_hits() == o.guard_hits(), tag
    return stats


@pytest.mark.parametrize("name,seq,m_lo,m_hi", ALL, ids=[c[0] for c in ALL])
def test_device_merge_matches_oracle(device_ranges, name, seq, m_lo, m_hi):
    _check(seq, m_lo, m_hi, name)


def test_a_list_head_write_that_only_mattered_inside_its_own_range(device_ranges):
    """tests/test_parallel_merge.py has the story (fuzz seed 430991): here the lane that merges the first range logs the write."""
    seq, m_lo, m_hi = fuzz_case(430991)
    _check(seq, m_lo, m_hi, "seed 430991")


def test_device_merge_on_fuzz_records_and_that_it_ran(device_ranges):
    on_device = left_to_host = 0
    for seed in range(60):
        seq, m_lo, m_hi = fuzz_case(41000 + seed, scale=3)
        dev, bailed, on_host, ranges, _ = _check(seq, m_lo, m_hi, f"fuzz {seed}")
        assert dev + bailed + on_host == ranges or dev == bailed == on_host == 0, (seed, dev, bailed, on_host, ranges)
        on_device += dev
        left_to_host += bailed
    assert on_device > 20 and left_to_host * 5 <= on_device, (on_device, left_to_host)
