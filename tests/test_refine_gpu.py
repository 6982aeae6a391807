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


def _seed_table_checks(sc, seq):
    """head/records of small_motifs.hip are self-consistent, and refinement takes exactly those seeds from the table"""
    head, rec = sc.small_motifs()
    d = sc.dispatch_seeds()
    assert len(head) == len(d)
    small = d["mlen"] <= 10
    assert (head[~small, 3] == -1).all()                       # long motifs never get a device result
    served = head[:, 3] == 0
    for i in np.nonzero(served)[0]:
        first, n_early, n_cls = (int(x) for x in head[i, :3])
        assert n_cls >= 0 and first + n_early + n_cls <= len(rec)      # 0 or 1: at most one class will be reported at the seed's end
        finals = rec[first + n_early:first + n_early + n_cls]
        assert len(set(int(c) for c in finals[:, 0])) == n_cls  # one record per rotation class
    before = ribbit_amd.small_motif_counters()
    sc.refine_bed("x")
    after = ribbit_amd.small_motif_counters()
    assert after[0] - before[0] == int(served.sum())
    assert after[1] - before[1] == int((head[small, 3] > 0).sum())
    return head


def test_small_motif_seeds_are_served_from_the_gpu_table():
    """... and the few seeds that show more rotation classes than a wavefront has lanes (long impure ones) are flagged and
    computed by the host twin; the BED text equals the oracle's either way"""
    name, seq, m_lo, m_hi = simulated_cases()[1]
    with ribbit_amd.Scanner(m_lo, m_hi) as sc, Oracle(seq, m_lo, m_hi) as o:
        sc.load_record(seq)
        head = _seed_table_checks(sc, seq)
        assert int((head[:, 3] == 0).sum()) > 1000 and 0 < int((head[:, 3] == 1).sum()) < 0.02 * len(head)
        o.run_all()
        assert sc.refine_bed("s") == o.refine_bed("s")


def test_substitution_rich_seed_of_a_ten_base_motif():
    """a 10-base repeat with substitutions in two units of three: dozens of rotation classes in one seed, early reports
    and restarts -- jobs and BED text from the GPU's table equal the oracle's"""
    rs = np.random.RandomState(5)
    unit = b"ACGTTGCAGC"
    units = []
    for k in range(90):
        u = bytearray(unit)
        if k % 3:
            p = int(rs.randint(0, 10))
            u[p] = ord(rs.choice([c for c in "ACGT" if c != chr(u[p])]))
        units.append(bytes(u))
    flank = bytes(rs.choice(list(b"ACGT"), 300).astype(np.uint8))
    seq = flank + b"".join(units) + flank[::-1] + b"AC" * 40 + flank + b"".join(units[:30]) + flank
    with ribbit_amd.Scanner(2, 12) as sc, Oracle(seq, 2, 12) as o:
        sc.load_record(seq)
        head = _seed_table_checks(sc, seq)
        assert (head[:, 3] == 0).any() and int(head[head[:, 3] == 0, 2].max()) >= 2     # a seed whose survivors' order matters
        o.run_all()
        wjobs, wpool = o.refine_jobs()
        gjobs, gpool = sc.refine_jobs()
        for f in ("seed_index", "atomicity", "query_start", "query_length", "ppr_length", "small"):
            assert np.array_equal(gjobs[f], wjobs[f]), f
        assert sc.refine_bed("s") == o.refine_bed("s")


@pytest.mark.gpu
def test_bed_view_is_the_same_text_without_the_copy():
    """Scanner.refine_bed_view: the library's buffer as a uint8 view (what bench.py's whole-path leg times, and what ribbit-hip
    writes to its output file) holds exactly the text refine_bed returns as a str"""
    from ribbit_amd.simulate import simulate_sequence
    seq, _ = simulate_sequence(300_000, 11, 2, 40)
    with ribbit_amd.Scanner(2, 40) as sc:
        sc.load_record(seq)
        text = sc.refine_bed("v")
        view = sc.refine_bed_view("v")
        assert view.dtype == np.uint8 and view.tobytes() == text.encode() and text.count("\n") > 100
    with ribbit_amd.Scanner(2, 6) as sc:
        sc.load_record(b"ACGTTGCA" * 3)
        assert len(sc.refine_bed_view("none")) == len(sc.refine_bed("none"))


def test_a_second_handle_refines_a_slice_of_the_dispatch_list():
    """ribbit_hip_adopt_dispatch: refinement of one record over several GPUs.  A second handle with the same record loaded takes
    slices of the dispatch list the first handle made, builds the composed planes on its own device (no scan, no merge) and
    refines them; the slices' texts back to back are the record's BED -- the oracle's -- however the list is cut, also when the
    first handle refines a slice of its own list."""
    from ribbit_amd.simulate import simulate_sequence
    seq, _ = simulate_sequence(400_000, 71, 2, 120, n_block_rate=0.2, lower_rate=0.1)
    with Oracle(seq, 2, 120) as o:
        o.run_all()
        want = o.refine_bed("rec")
    with ribbit_amd.Scanner(2, 120) as a, ribbit_amd.Scanner(2, 120) as b:
        a.load_record(seq)
        d = a.dispatch_seeds().copy()
        assert len(d) > 5000
        b.load_record(seq)
        for cuts in ([0, len(d)], [0, len(d) // 3, len(d)], [0, 1, 17, len(d) // 2, len(d) - 1, len(d)]):
            parts = []
            for k, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
                h = a if k % 2 == 0 and len(cuts) > 2 else b          # the first handle takes slices of its own list too
                h.adopt_dispatch(d[lo:hi])
                parts.append(h.refine_bed("rec"))
                assert not h.refine_met_empty_query()
            assert "".join(parts) == want, cuts
