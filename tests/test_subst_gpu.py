"""GPU parity tests of the substitution stage (window-scan kernel + host state machine + merges)
against the CPU oracle and the committed fixtures.  Bit-exact."""
import glob
import os

import numpy as np
import pytest

import ribbit_amd
from cases import edge_cases, simulated_cases
from oracle_lib import LIST_PERFECT, LIST_SUBST, Oracle

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL = edge_cases() + simulated_cases()


@pytest.mark.parametrize("name,seq,m_lo,m_hi", ALL, ids=[c[0] for c in ALL])
def test_subst_stage_matches_oracle(name, seq, m_lo, m_hi):
    with ribbit_amd.Scanner(m_lo, m_hi) as sc, Oracle(seq, m_lo, m_hi) as o:
        sc.load_record(seq)
        o.run_perfect()
        o.run_subst()
        assert np.array_equal(sc.subst_calls().view("<i4"), o.calls(LIST_SUBST).view("<i4"))
        perfect, subst = sc.processShiftXORswithSubstitutions()
        assert np.array_equal(perfect.view("<i4"), o.seeds(LIST_PERFECT).view("<i4"))
        assert np.array_equal(subst.view("<i4"), o.seeds(LIST_SUBST).view("<i4"))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))), ids=lambda p: os.path.basename(p)[:-4])
def test_subst_stage_matches_fixture(path):
    g = np.load(path)
    with ribbit_amd.Scanner(int(g["m_lo"]), int(g["m_hi"])) as sc:
        sc.load_record(g["seq"].tobytes())
        assert np.array_equal(sc.subst_calls().view("<i4"), g["subst_calls"].view("<i4"))
        perfect, subst = sc.processShiftXORswithSubstitutions()
        assert np.array_equal(perfect.view("<i4"), g["perfect_after_s"].view("<i4"))
        assert np.array_equal(subst.view("<i4"), g["subst_after_s"].view("<i4"))


def test_stage_order_is_enforced():
    name, seq, m_lo, m_hi = simulated_cases()[0]
    with ribbit_amd.Scanner(m_lo, m_hi) as sc:
        sc.load_record(seq)
        sc.processShiftXORswithSubstitutions()
        with pytest.raises(ribbit_amd.RibbitHipError):
            sc.processShiftXORsPerfect()      # the perfect list has been re-typed by the later stage
        sc.load_record(seq)
        sc.processShiftXORsPerfect()
