"""GPU parity tests (run with -m gpu on an MI355X): the HIP path through the C ABI against the
CPU oracle on the same seeded inputs and against the committed fixtures.  Bit-exact (integer work).
"""
import glob
import os

import numpy as np
import pytest

import pyref
import ribbit_amd
from cases import edge_cases, simulated_cases
from oracle_lib import LIST_PERFECT, Oracle

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL = edge_cases() + simulated_cases()


def _unpack(words, n):
    bits = np.unpackbits(words.view(np.uint8), bitorder="little")
    return bits[:n]


def _expected_runs(o, m_lo, m_hi):
    """all maximal runs of X_m & ~N with length >= min(c1, 32), from the oracle's planes"""
    nmask = o.nmask()
    L = len(nmask)
    out = []
    for m in range(m_lo, m_hi + 1):
        sp = min(12 - m if m <= 6 else m, 32)
        y = o.plane(m) & (1 - nmask)
        for s, e in pyref.runs_of_ones(y):
            if e - s >= sp:
                term = 2 if e == L else (1 if nmask[e] else 0)
                out.append((s, e, m, term))
    return out


@pytest.mark.parametrize("name,seq,m_lo,m_hi", ALL, ids=[c[0] for c in ALL])
def test_pack_and_planes_match_oracle(name, seq, m_lo, m_hi):
    with ribbit_amd.Scanner(m_lo, m_hi) as sc, Oracle(seq, m_lo, m_hi) as o:
        sc.load_record(seq)
        n = len(seq)
        code, nmask = o.codes(), o.nmask()
        assert np.array_equal(_unpack(sc.packed_plane(0), n), code >> 1)
        assert np.array_equal(_unpack(sc.packed_plane(1), n), code & 1)
        assert np.array_equal(_unpack(sc.packed_plane(2), n), nmask)
        # positions >= L in the last word must break runs
        assert _unpack(sc.packed_plane(2), (n // 32 + 1) * 32)[n:].all()
        for s in sorted({sc.min_shift, m_lo, (m_lo + m_hi) // 2, m_hi, sc.max_shift}):
            assert np.array_equal(sc.plane_bits(s), o.plane(s)), f"shift {s}"
            if n > 10:
                a, b = n // 7, n - n // 5
                assert sc.range_popcount(s, a, b) == o.range_count(s, a, b)
                assert np.array_equal(sc.plane_bits(s, a, b), o.plane(s)[a:b])


@pytest.mark.parametrize("name,seq,m_lo,m_hi", ALL, ids=[c[0] for c in ALL])
def test_perfect_scan_matches_oracle(name, seq, m_lo, m_hi):
    with ribbit_amd.Scanner(m_lo, m_hi) as sc, Oracle(seq, m_lo, m_hi) as o:
        sc.load_record(seq)
        runs = sc.scan_perfect_runs()
        got = [tuple(int(x) for x in r) for r in runs]
        assert got == sorted(_expected_runs(o, m_lo, m_hi), key=lambda t: (t[2], t[0]))
        o.run_perfect()
        assert np.array_equal(sc.perfect_calls(), o.calls(LIST_PERFECT).astype(ribbit_amd.CALL_DT))
        assert np.array_equal(sc.processShiftXORsPerfect(), o.seeds(LIST_PERFECT).astype(ribbit_amd.SEED_DT))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))),
                         ids=lambda p: os.path.basename(p)[:-4])
def test_perfect_stage_matches_fixture(path):
    g = np.load(path)
    with ribbit_amd.Scanner(int(g["m_lo"]), int(g["m_hi"])) as sc:
        sc.load_record(g["seq"].tobytes())
        assert np.array_equal(sc.perfect_calls().view("<i4"), g["perfect_calls"].view("<i4"))
        assert np.array_equal(sc.processShiftXORsPerfect().view("<i4"), g["perfect_after_p"].view("<i4"))


def test_reload_and_rescan_is_idempotent():
    name, seq, m_lo, m_hi = simulated_cases()[1]
    with ribbit_amd.Scanner(m_lo, m_hi) as sc:
        sc.load_record(seq)
        a = sc.scan_perfect_runs()
        b = sc.scan_perfect_runs()
        sc.load_record(edge_cases()[5][1])
        sc.scan_perfect_runs()
        sc.load_record(seq)
        c = sc.scan_perfect_runs()
        assert np.array_equal(a, b) and np.array_equal(a, c)


def test_full_size_properties_100mbp():
    """BASELINE.json config 2 size (100 Mbp, m=2..100): size-independent properties, no oracle."""
    from ribbit_amd.simulate import simulate_sequence
    total = 100_000_000
    unit, _ = simulate_sequence(1_000_000, 2, 2, 100)
    seq = unit * (total // len(unit))
    with ribbit_amd.Scanner(2, 100) as sc:
        sc.load_record(seq)
        runs = sc.scan_perfect_runs()
        one = None
        with ribbit_amd.Scanner(2, 100) as sc1:
            sc1.load_record(unit)
            one = sc1.scan_perfect_runs()
    # runs are sorted by (mlen, start), disjoint per motif, and inside the record
    key = runs["mlen"].astype(np.int64) << 32 | runs["start"]
    assert np.all(np.diff(key) > 0)
    assert runs["start"].min() >= 0 and runs["end"].max() <= total
    same = runs["mlen"][1:] == runs["mlen"][:-1]
    assert np.all(runs["start"][1:][same] > runs["end"][:-1][same])
    # periodic input: every run of the 1-Mbp unit that stays clear of the unit's edges recurs in every copy
    m = 1_000_000
    inner = one[(one["start"] > 200) & (one["end"] < m - 200)]
    for copy in (0, 37, 99):
        sel = runs[(runs["start"] >= copy * m + 200) & (runs["end"] < (copy + 1) * m - 200) & (runs["start"] > copy * m + 200)]
        sel = sel.copy(); sel["start"] -= copy * m; sel["end"] -= copy * m
        assert np.array_equal(np.sort(sel, order=["mlen", "start"]), np.sort(inner, order=["mlen", "start"]))


def test_begin_end_on_two_handles_interleaved_gives_the_same_runs():
    """ribbit_hip_scan_perfect_begin/_end: two records in flight on two handles (own streams), finished out of
    phase, must give what the one-call scan gives; _end without _begin is an error."""
    (_, seq_a, m_lo, m_hi), (_, seq_b, _, _) = simulated_cases()[1], simulated_cases()[3]
    with ribbit_amd.Scanner(m_lo, m_hi) as a, ribbit_amd.Scanner(m_lo, m_hi) as b:
        a.load_record(seq_a); want_a = a.scan_perfect_runs()
        b.load_record(seq_b); want_b = b.scan_perfect_runs()
        for _ in range(3):
            a.load_record(seq_a); a.scan_perfect_begin()
            b.load_record(seq_b); b.scan_perfect_begin()
            got_a, halves = a.scan_perfect_end()
            assert len(halves) == 0 and np.array_equal(got_a.view("<i4"), want_a.view("<i4"))
            a.load_record(seq_b); a.scan_perfect_begin()          # next batch enqueued before b is collected
            got_b, _ = b.scan_perfect_end()
            assert np.array_equal(got_b.view("<i4"), want_b.view("<i4"))
            got_a2, _ = a.scan_perfect_end()
            assert np.array_equal(got_a2.view("<i4"), want_b.view("<i4"))
        with pytest.raises(ribbit_amd.RibbitHipError, match="no perfect scan in flight"):
            a.scan_perfect_end()
        # the later stages still work on a handle that used the split calls
        a.load_record(seq_a)
        assert len(a.processShiftXORsPerfect()) > 0


@pytest.mark.parametrize("n", [1, 31, 32, 33, 4099, 70_001])
def test_pack_of_every_byte_value_matches_oracle(n):
    """The encode (fasta_utils.cpp:94-114) on arbitrary bytes: A/C/G/T in either case, everything else -- control
    characters, digits, other letters, bytes >= 128 -- is N.  Odd lengths exercise the unaligned tail."""
    rs = np.random.RandomState(n)
    raw = rs.randint(1, 256, size=n).astype(np.uint8)           # no NUL: the record is a C string for the oracle
    mix = np.where(rs.rand(n) < 0.6, np.frombuffer(b"ACGTacgt", dtype=np.uint8)[rs.randint(0, 8, size=n)], raw)
    seq = mix.astype(np.uint8).tobytes()
    with ribbit_amd.Scanner(2, 8) as sc, Oracle(seq, 2, 8) as o:
        sc.load_record(seq)
        code, nmask = o.codes(), o.nmask()
        assert np.array_equal(_unpack(sc.packed_plane(0), n), code >> 1)
        assert np.array_equal(_unpack(sc.packed_plane(1), n), code & 1)
        assert np.array_equal(_unpack(sc.packed_plane(2), n), nmask)
        assert _unpack(sc.packed_plane(2), (n // 32 + 1) * 32)[n:].all()


@pytest.mark.parametrize("seed,bases,m_lo,m_hi", [(71, 3_000_000, 2, 100), (72, 2_000_000, 5, 40), (73, 1_500_000, 30, 200), (74, 2_000_000, 2, 14)])
def test_perfect_calls_on_megabase_records_match_oracle(seed, bases, m_lo, m_hi):
    """Hundreds of tiles per record: the candidate queue of the perfect scan fills and flushes across motifs, pairs
    run in place when dense, runs cross tile edges -- all against the oracle's perfect stage (calls and seeds)."""
    from ribbit_amd.simulate import simulate_sequence
    seq, _ = simulate_sequence(bases, seed, m_lo, min(m_hi, 100), n_block_rate=0.2)
    with ribbit_amd.Scanner(m_lo, m_hi) as sc, Oracle(seq, m_lo, m_hi) as o:
        sc.load_record(seq)
        o.run_perfect()
        assert np.array_equal(sc.perfect_calls(), o.calls(LIST_PERFECT).astype(ribbit_amd.CALL_DT))
        assert np.array_equal(sc.processShiftXORsPerfect(), o.seeds(LIST_PERFECT).astype(ribbit_amd.SEED_DT))


def test_event_buffer_overflow_is_retried_with_a_larger_buffer():
    """First guess of the event capacity far too small (64 events per region): every stage must notice the
    overflow, size its regions for the fullest one and come out with the same lists."""
    name, seq, m_lo, m_hi = simulated_cases()[1]
    with ribbit_amd.Scanner(m_lo, m_hi) as ref, ribbit_amd.Scanner(m_lo, m_hi) as sc:
        ref.load_record(seq)
        want_runs = ref.scan_perfect_runs()
        want = ref.processShiftXORsAnchored()
        sc.debug_set_event_capacity(64 * 64)
        sc.load_record(seq)
        assert np.array_equal(sc.scan_perfect_runs().view("<i4"), want_runs.view("<i4"))
        sc.load_record(seq); sc.scan_perfect_begin()
        assert np.array_equal(sc.scan_perfect_end()[0].view("<i4"), want_runs.view("<i4"))
        got = sc.processShiftXORsAnchored()
        for a, b in zip(got, want):
            assert np.array_equal(a.view("<i4"), b.view("<i4"))


@pytest.mark.parametrize("length", [16383, 16384, 16385, 32767, 32768, 32769, 15871, 15872, 15873, 8 * 16384])
def test_record_lengths_at_tile_boundaries_with_a_run_open_at_the_end(length):
    """The end-of-sequence event sits at position L: exactly on, just before and just after a tile boundary of the
    perfect / window kernels (16384 bases) and of the anchored kernel (15872), with a repeat still running there."""
    rs = np.random.RandomState(length)
    body = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rs.randint(0, 4, size=length - 90)])
    seq = body + (b"CAT" * 30)
    assert len(seq) == length
    with ribbit_amd.Scanner(2, 20) as sc, Oracle(seq, 2, 20) as o:
        sc.load_record(seq)
        o.run_all()
        assert np.array_equal(sc.perfect_calls().view("<i4"), o.calls(LIST_PERFECT).view("<i4"))
        perfect, subst, anchored = sc.processShiftXORsAnchored()
        assert np.array_equal(perfect.view("<i4"), o.seeds(LIST_PERFECT).view("<i4"))
        assert np.array_equal(sc.dispatch_seeds().view("<i4"), o.dispatch().view("<i4"))
