"""GPU fuzz: the whole path (pack, three scan kernels, pairing kernels, state machines, merges, refinement kernels,
SSW, BED) against the oracle pipeline on seeded adversarial records (tests/fuzz.py), motif ranges up to 990.
One Scanner per motif range; any disagreement prints the seed."""
import numpy as np
import pytest

import ribbit_amd
from fuzz import fuzz_case
from oracle_lib import LIST_ANCHORED, LIST_PERFECT, LIST_SUBST, Oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("block", range(5))
def test_whole_path_matches_oracle_on_fuzzed_records(block):
    for seed in range(31000 + 40 * block, 31000 + 40 * (block + 1)):
        seq, m_lo, m_hi = fuzz_case(seed)
        tag = f"seed {seed}: {len(seq)} bases, -m {m_lo} -M {m_hi}"
        with ribbit_amd.Scanner(m_lo, m_hi) as sc, Oracle(seq, m_lo, m_hi) as o:
            sc.load_record(seq)
            o.run_all()
            assert np.array_equal(sc.perfect_calls().view("<i4"), o.calls(LIST_PERFECT).view("<i4")), tag
            assert np.array_equal(sc.subst_calls().view("<i4"), o.calls(LIST_SUBST).view("<i4")), tag
            assert np.array_equal(sc.anchored_calls().view("<i4"), o.calls(LIST_ANCHORED).view("<i4")), tag
            perfect, subst, anchored = sc.processShiftXORsAnchored()
            assert np.array_equal(perfect.view("<i4"), o.seeds(LIST_PERFECT).view("<i4")), tag
            assert np.array_equal(subst.view("<i4"), o.seeds(LIST_SUBST).view("<i4")), tag
            assert np.array_equal(anchored.view("<i4"), o.seeds(LIST_ANCHORED).view("<i4")), tag
            assert np.array_equal(sc.dispatch_seeds().view("<i4"), o.dispatch().view("<i4")), tag
            assert sc.guard_hits() == o.guard_hits(), tag
            assert sc.refine_bed("fz").split("\n") == o.refine_bed("fz").split("\n"), tag
