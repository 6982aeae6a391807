"""CPU tests of the refinement front half (rows a13-a15): the product's host implementation
(ribbit_amd/csrc/refine.cpp, through ribbit_host_refine_jobs) against the oracle's restatement
(oracle/ribbit_oracle_refine.cpp) on the oracle's dispatch lists and composed planes."""
import ctypes as C

import numpy as np
import pytest

import ribbit_amd
from cases import edge_cases, simulated_cases
from oracle_lib import Oracle, RefineParams, lib

ALL = edge_cases() + simulated_cases()


def _same_jobs(got, gpool, want, wpool):
    assert len(got) == len(want)
    for f in ("seed_index", "seed_type", "motif_length", "atomicity", "query_start", "query_length", "ppr_length", "small"):
        assert np.array_equal(got[f], want[f]), f
    gm = [m for _, m in ribbit_amd._jobs_with_motifs(got, gpool)]
    wm = [m for _, m in ribbit_amd._jobs_with_motifs(want, wpool)]
    assert gm == wm


@pytest.mark.parametrize("name,seq,m_lo,m_hi", ALL, ids=[c[0] for c in ALL])
def test_refine_jobs_match_oracle(name, seq, m_lo, m_hi):
    with Oracle(seq, m_lo, m_hi) as o:
        o.run_all()
        want, wpool = o.refine_jobs()
        xa, stride = ribbit_amd.pack_bit_planes([o.plane(m) for m in range(m_lo, m_hi + 1)], len(seq))
        got, gpool = ribbit_amd.host_refine_jobs(m_lo, m_hi, seq, xa, stride, o.dispatch())
    _same_jobs(got, gpool, want, wpool)


@pytest.mark.parametrize("slices", [1, 3, 7])
def test_the_pipelines_slice_builder_makes_the_same_jobs(slices):
    """The GPU refinement pipeline sets its jobs up with build_align_jobs_slices (all slices of the seed list in one parallel
    region, a slice assembled and handed over by the thread that finished its last chunk); the host-only entry point uses the
    plain builder.  RIBBIT_DEBUG_JOB_SLICES routes the host entry point through the pipeline's builder: the same jobs, the same
    motif strings, in seed order, whatever the number of slices and threads -- checked against the oracle's jobs on the larger
    simulated records (a few thousand seeds each, so that slices hold several chunks' worth and several threads take part)."""
    import os
    cases = [c for c in ALL if len(c[1]) >= 100_000][:4]
    assert cases
    for name, seq, m_lo, m_hi in cases:
        with Oracle(seq, m_lo, m_hi) as o:
            o.run_all()
            want, wpool = o.refine_jobs()
            xa, stride = ribbit_amd.pack_bit_planes([o.plane(m) for m in range(m_lo, m_hi + 1)], len(seq))
            os.environ["RIBBIT_DEBUG_JOB_SLICES"] = str(slices)
            os.environ["RIBBIT_THREADS"] = "5"
            try:
                got, gpool = ribbit_amd.host_refine_jobs(m_lo, m_hi, seq, xa, stride, o.dispatch())
            finally:
                del os.environ["RIBBIT_DEBUG_JOB_SLICES"], os.environ["RIBBIT_THREADS"]
        _same_jobs(got, gpool, want, wpool)


def test_defaults_follow_ribbit_cpp():
    for m_lo, m_hi in ((2, 100), (5, 40), (3, 9), (30, 100)):
        a = RefineParams()
        lib().rbo_refine_params_default(C.byref(a), m_lo, m_hi)
        b = ribbit_amd.RefineParams()
        ribbit_amd.load_library().ribbit_refine_params_default(C.byref(b), m_lo, m_hi)
        assert list(a.min_length) == list(b.min_length) and list(a.perfect_units) == list(b.perfect_units)
        assert a.purity_threshold == b.purity_threshold and a.continuous_ones_threshold == b.continuous_ones_threshold
    # ribbit.cpp:153-159: 12, or 2*m when that is larger; :166-173: 8/4/3/2; :219-235: factors inherit
    p = ribbit_amd.RefineParams()
    ribbit_amd.load_library().ribbit_refine_params_default(C.byref(p), 5, 40)
    assert p.min_length[5] == 12 and p.min_length[7] == 14 and p.min_length[40] == 80
    assert p.min_length[1] == 12 and p.min_length[2] == 12 and p.min_length[4] == 16 and p.min_length[3] == 12
    assert [p.perfect_units[k] for k in (1, 2, 3, 4, 40)] == [8, 4, 3, 2, 2]


def test_jobs_are_plausible_on_simulated_repeats():
    # sanity (not parity): most planted small-motif loci produce a job whose motif is a rotation of the planted one
    from ribbit_amd.simulate import simulate_sequence
    seq, truth = simulate_sequence(120_000, 9, 2, 8)
    with Oracle(seq, 2, 8) as o:
        o.run_all()
        jobs, pool = o.refine_jobs()
    motifs = ribbit_amd._jobs_with_motifs(jobs, pool)
    hits = 0
    for (ts, te, m, motif) in truth:
        rots = {motif[i:] + motif[:i] for i in range(m)}
        hits += any(int(j["query_start"]) < te and int(j["query_start"]) + int(j["query_length"]) > ts and mot in rots for j, mot in motifs)
    assert hits >= 0.85 * len(truth), (hits, len(truth))


def test_consensus_rows_with_every_simd_form_give_the_same_bed():
    """consensus_row (mostFrequentLongerMotif, parse_seed.cpp:153-256) compares 8 symbols at a time in plain C++, 32 with AVX2,
    64 with AVX-512BW (chosen at run time; RIBBIT_HOST_SIMD caps the choice and is read once per process, hence the
    subprocesses): the BED of a record with long motifs -- where the rows decide the motif strings -- must not depend on it,
    and must be the oracle's."""
    import hashlib
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = (
        "import sys, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
        "import ribbit_amd\n"
        "from oracle_lib import Oracle\n"
        "from ribbit_amd.simulate import simulate_sequence\n"
        "seq, _ = simulate_sequence(250000, 91, 2, 300, n_block_rate=0.2)\n"
        "with Oracle(seq, 2, 300) as o:\n"
        "    o.run_all(); d = o.dispatch()\n"
        "    xa, stride = ribbit_amd.pack_bit_planes([o.plane(m) for m in range(2, 301)], len(seq))\n"
        "    want = o.refine_bed('x')\n"
        "got = ribbit_amd.host_refine_bed(2, 300, seq, xa, stride, d, 'x')\n"
        "print(hashlib.sha256(got.encode()).hexdigest(), got == want, got.count(chr(10)))\n" % (root, root))
    seen = set()
    for level in ("0", "1", "2"):
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600, env=dict(os.environ, RIBBIT_HOST_SIMD=level))
        assert r.returncode == 0, r.stderr[-1500:]
        digest, same, rows = r.stdout.split()
        assert same == "True" and int(rows) > 500, r.stdout
        seen.add(digest)
    assert len(seen) == 1


def test_the_recursion_cut_into_nodes_pieces_and_levels_gives_the_same_bed():
    """processSeed's recursion on the flanks (parse_seed.cpp:443-463) as the GPU path cuts it -- nodes put off from a threshold
    on, a level at a time, their rows in pieces labelled with their place in the seed's recursion tree and sorted into
    pre-order at the end (refine.h: DeferredNode) -- run here by the host-only entry point (RIBBIT_HOST_DEFER: every level is
    refined by host code; every third level finishes its subtrees by recursion, like the GPU path's last level): the BED is
    the oracle's whatever the threshold, and the counters show that nodes were put off over several levels."""
    import os
    from fuzz import fuzz_case
    from ribbit_amd.simulate import simulate_sequence
    cases = [(simulate_sequence(250_000, 91, 2, 300, n_block_rate=0.2)[0], 2, 300), (simulate_sequence(150_000, 7, 2, 500, lower_rate=0.2)[0], 2, 500)]
    cases += [fuzz_case(s) for s in (9001, 9007, 9013, 82531)] + [c[1:] for c in ALL[:6]]
    os.environ["RIBBIT_HOST_DEFER"] = "1"
    try:
        deep = 0
        for seq, m_lo, m_hi in cases:
            with Oracle(seq, m_lo, m_hi) as o:
                o.run_all()
                d = o.dispatch()
                xa, stride = ribbit_amd.pack_bit_planes([o.plane(m) for m in range(m_lo, m_hi + 1)], len(seq))
                want = o.refine_bed("x")
            for threshold in ("1", "60", "700"):
                os.environ["RIBBIT_DEFER_MIN"] = threshold
                before = ribbit_amd.level_counters()
                got = ribbit_amd.host_refine_bed(m_lo, m_hi, seq, xa, stride, d, "x")
                levels, nodes, _ = (b - a for a, b in zip(before, ribbit_amd.level_counters()))
                assert got == want, (len(seq), m_lo, m_hi, threshold)
                deep = max(deep, levels)
                if threshold == "1" and len(seq) >= 150_000:
                    assert nodes > 100 and levels >= 2, (levels, nodes)
        assert deep >= 3
    finally:
        os.environ.pop("RIBBIT_HOST_DEFER", None)
        os.environ.pop("RIBBIT_DEFER_MIN", None)
