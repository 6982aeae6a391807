"""GPU parity tests of rows a13-a15 through the handle: the batched longestContinuousMatches kernel
and the alignment-job list, against the oracle.  Bit-exact."""
import numpy as np
import pytest

import pyref
import ribbit_amd
from cases import edge_cases, simulated_cases
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu
ALL = edge_cases() + simulated_cases()


@pytest.mark.parametrize("name,seq,m_lo,m_hi", ALL, ids=[c[0] for c in ALL])
def test_longest_runs_and_jobs_match_oracle(name, seq, m_lo, m_hi):
    with ribbit_amd.Scanner(m_lo, m_hi) as sc, Oracle(seq, m_lo, m_hi) as o:
        sc.load_record(seq)
        o.run_all()
        d = o.dispatch()
        got = sc.seed_longest_runs()
        assert len(got) == len(d)
        planes = {}
        for i, s in enumerate(d):
            m = int(s["mlen"])
            if m not in planes:
                planes[m] = o.plane(m)
            runs = pyref.runs_of_ones(planes[m][int(s["start"]):int(s["end"])])
            want = max((e - a for a, e in runs), default=0)
            assert int(got[i]) == want, (i, s)
        wjobs, wpool = o.refine_jobs()
        gjobs, gpool = sc.refine_jobs()
        assert len(gjobs) == len(wjobs)
        for f in ("seed_index", "seed_type", "motif_length", "atomicity", "query_start", "query_length", "ppr_length", "small"):
            assert np.array_equal(gjobs[f], wjobs[f]), f
        assert [m for _, m in ribbit_amd._jobs_with_motifs(gjobs, gpool)] == [m for _, m in ribbit_amd._jobs_with_motifs(wjobs, wpool)]
