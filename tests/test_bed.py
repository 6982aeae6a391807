"""Rows f1/f4 on CPU: the product's BED rows (own refinement scans, own SSW, own CIGAR processing:
ribbit_host_refine_bed) against the oracle's (restated refinement around the REFERENCE's own SSW from
oracle/_ref) on the oracle's dispatch lists.  Text-identical, purity column included."""
import os

import pytest

import ribbit_amd
from cases import edge_cases, large_motif_cases, simulated_cases
from oracle_lib import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libssw_ref.so")),
                                reason="oracle/_ref/libssw_ref.so not built")
ALL = edge_cases() + simulated_cases() + large_motif_cases()


@pytest.mark.parametrize("name,seq,m_lo,m_hi", ALL, ids=[c[0] for c in ALL])
def test_bed_rows_match_oracle(name, seq, m_lo, m_hi):
    with Oracle(seq, m_lo, m_hi) as o:
        o.run_all()
        want = o.refine_bed(name)
        xa, stride = ribbit_amd.pack_bit_planes([o.plane(m) for m in range(m_lo, m_hi + 1)], len(seq))
        got = ribbit_amd.host_refine_bed(m_lo, m_hi, seq, xa, stride, o.dispatch(), name)
    assert got.split("\n") == want.split("\n")


def test_bed_has_eleven_columns_and_sane_values():
    name, seq, m_lo, m_hi = simulated_cases()[0]
    with Oracle(seq, m_lo, m_hi) as o:
        o.run_all()
        bed = o.refine_bed("chrSim")
    rows = [r.split("\t") for r in bed.strip().split("\n")]
    assert len(rows) > 100
    for r in rows:
        assert len(r) == 11 and r[0] == "chrSim" and r[8] == "+" and r[9].startswith("SEED-")
        start, end = int(r[1]), int(r[2])
        assert 0 <= start < end <= len(seq) and int(r[5]) == end - start
        assert 0.0 < float(r[7]) <= 1.0
