"""CPU fuzz: the oracle's call logs replayed through the product's independently written merges, refinement and
BED writer on seeded adversarial records (tests/fuzz.py).  Any disagreement prints the seed."""
import os

import numpy as np
import pytest

import ribbit_amd
from fuzz import fuzz_case
from oracle_lib import LIST_ANCHORED, LIST_PERFECT, LIST_SUBST, Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_REF_SSW = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libssw_ref.so"))


def _seeds(block):
    if block < 6:
        return [(seed, 1) for seed in range(9000 + 40 * block, 9000 + 40 * (block + 1))]
    # records stretched over several kernel tiles; 82531 is the case that exposed an out-of-bounds read in the
    # atomicity test of motifs longer than 192 bases
    return [(82531, 16)] + [(seed, 16) for seed in range(86000 + 6 * (block - 6), 86000 + 6 * (block - 5))]


@pytest.mark.parametrize("block", range(8))
def test_merges_and_bed_rows_agree_with_the_oracle_on_fuzzed_records(block):
    for seed, scale in _seeds(block):
        seq, m_lo, m_hi = fuzz_case(seed, scale)
        tag = f"seed {seed}: {len(seq)} bases, -m {m_lo} -M {m_hi}"
        with Oracle(seq, m_lo, m_hi) as o:
            o.run_perfect(); pc = o.calls(LIST_PERFECT)
            o.run_subst(); sc = o.calls(LIST_SUBST)
            o.run_anchor_planes()
            xa, stride = ribbit_amd.pack_bit_planes([o.plane(m) for m in range(m_lo, m_hi + 1)], len(seq))
            o.run_anchored(); o.run_dispatch()
            r = ribbit_amd.host_replay_calls(m_lo, m_hi, seq, pc, sc, o.calls(LIST_ANCHORED), xa, stride)
            assert np.array_equal(r["perfect"].view("<i4"), o.seeds(LIST_PERFECT).view("<i4")), tag
            assert np.array_equal(r["subst"].view("<i4"), o.seeds(LIST_SUBST).view("<i4")), tag
            assert np.array_equal(r["anchored"].view("<i4"), o.seeds(LIST_ANCHORED).view("<i4")), tag
            assert np.array_equal(r["dispatch"].view("<i4"), o.dispatch().view("<i4")), tag
            assert r["guard_hits"] == o.guard_hits(), tag
            if HAVE_REF_SSW and len(seq):
                got = ribbit_amd.host_refine_bed(m_lo, m_hi, seq, xa, stride, o.dispatch(), "fz")
                assert got.split("\n") == o.refine_bed("fz").split("\n"), tag


def test_refinement_with_random_length_and_unit_tables_agrees_with_the_oracle():
    """MINIMUM_LENGTH / PERFECT_UNITS other than the defaults (what -l / --min-units / --perfect-units set)."""
    import ctypes as C
    from oracle_lib import RefineParams as OracleRefineParams, lib as oracle_lib_handle
    if not HAVE_REF_SSW:
        pytest.skip("oracle/_ref/libssw_ref.so not built")
    for seed in range(12000, 12040):
        seq, m_lo, m_hi = fuzz_case(seed)
        if len(seq) < 40:
            continue
        rs = np.random.RandomState(seed)
        style = rs.randint(0, 3)
        orp, prp = OracleRefineParams(), ribbit_amd.RefineParams()
        oracle_lib_handle().rbo_refine_params_default(C.byref(orp), m_lo, m_hi)
        ribbit_amd.load_library().ribbit_refine_params_default(C.byref(prp), m_lo, m_hi)
        for k in range(1024):
            ml = int(rs.randint(0, 40)) if style == 0 else k * int(rs.randint(1, 6)) if style == 1 else orp.min_length[k]
            pu = int(rs.randint(0, 6)) if style != 1 else orp.perfect_units[k]
            orp.min_length[k] = prp.min_length[k] = ml
            orp.perfect_units[k] = prp.perfect_units[k] = pu
        with Oracle(seq, m_lo, m_hi) as o:
            o.run_all()
            xa, stride = ribbit_amd.pack_bit_planes([o.plane(m) for m in range(m_lo, m_hi + 1)], len(seq))
            got = ribbit_amd.host_refine_bed(m_lo, m_hi, seq, xa, stride, o.dispatch(), "fz", prp)
            assert got.split("\n") == o.refine_bed("fz", orp).split("\n"), f"seed {seed} style {style}"
