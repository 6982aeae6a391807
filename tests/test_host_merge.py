"""CPU tests of the product's host-side seed-list merges (ribbit_amd/csrc/seed_lists.cpp) through
the C ABI entry ribbit_host_replay_calls: the oracle's call logs are replayed through the product's
own, independently written merges and the resulting lists must equal the oracle's bit for bit.
No GPU needed; the GPU tests check that the kernels reproduce the call logs themselves."""
import glob
import os

import numpy as np
import pytest

import ribbit_amd
from cases import edge_cases, large_motif_cases, simulated_cases
from oracle_lib import LIST_ANCHORED, LIST_PERFECT, LIST_SUBST, Oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL = edge_cases() + simulated_cases() + large_motif_cases()


def test_pack_planes_matches_oracle_encoding():
    for name, seq, m_lo, m_hi in edge_cases():
        hi, lo, brk = ribbit_amd.pack_planes(seq, m_hi)
        n = len(seq)
        with Oracle(seq, m_lo, m_hi) as o:
            code, nmask = o.codes(), o.nmask()
        bits = lambda w: np.unpackbits(w.view(np.uint8), bitorder="little")
        assert np.array_equal(bits(hi)[:n], code >> 1) and np.array_equal(bits(lo)[:n], code & 1), name
        assert np.array_equal(bits(brk)[:n], nmask) and bits(brk)[n:].all() and not bits(hi)[n:].any(), name


@pytest.mark.parametrize("name,seq,m_lo,m_hi", ALL, ids=[c[0] for c in ALL])
def test_replay_of_oracle_calls_reproduces_oracle_lists(name, seq, m_lo, m_hi):
    with Oracle(seq, m_lo, m_hi) as o:
        o.run_perfect()
        perfect_after_p = o.seeds(LIST_PERFECT)
        pcalls = o.calls(LIST_PERFECT)
        r = ribbit_amd.host_replay_calls(m_lo, m_hi, seq, pcalls)
        assert np.array_equal(r["perfect"].view("<i4"), perfect_after_p.view("<i4"))
        o.run_subst()
        scalls = o.calls(LIST_SUBST)
        r = ribbit_amd.host_replay_calls(m_lo, m_hi, seq, pcalls, scalls)
        assert np.array_equal(r["perfect"].view("<i4"), o.seeds(LIST_PERFECT).view("<i4"))
        assert np.array_equal(r["subst"].view("<i4"), o.seeds(LIST_SUBST).view("<i4"))
        assert r["guard_hits"] == 0
        # anchored stage: the oracle's composed planes and call log through the product's merges
        o.run_anchor_planes()
        xa, stride = ribbit_amd.pack_bit_planes([o.plane(m) for m in range(m_lo, m_hi + 1)], len(seq))
        o.run_anchored()
        o.run_dispatch()
        r = ribbit_amd.host_replay_calls(m_lo, m_hi, seq, pcalls, scalls, o.calls(LIST_ANCHORED), xa, stride)
        assert np.array_equal(r["perfect"].view("<i4"), o.seeds(LIST_PERFECT).view("<i4"))
        assert np.array_equal(r["subst"].view("<i4"), o.seeds(LIST_SUBST).view("<i4"))
        assert np.array_equal(r["anchored"].view("<i4"), o.seeds(LIST_ANCHORED).view("<i4"))
        assert np.array_equal(r["dispatch"].view("<i4"), o.dispatch().view("<i4"))
        assert r["guard_hits"] == o.guard_hits()
        # the same without being given the composed planes: the merges' range reads recompute the slice of XA_m they
        # need from the packed planes (what the GPU path does: the planes the kernel composed never leave HBM)
        r2 = ribbit_amd.host_replay_calls(m_lo, m_hi, seq, pcalls, scalls, o.calls(LIST_ANCHORED))
        for k in ("perfect", "subst", "anchored", "dispatch"):
            assert np.array_equal(r2[k].view("<i4"), r[k].view("<i4")), k
        assert r2["guard_hits"] == r["guard_hits"]


@pytest.mark.parametrize("name,seq,m_lo,m_hi", ALL, ids=[c[0] for c in ALL])
def test_recomputed_composed_plane_slices_equal_the_oracles_planes(name, seq, m_lo, m_hi):
    """HostPlanes::xa_slice against the oracle's composed planes (fasta_utils.cpp:143-161), through the longest-run scan of
    ribbit_host_refine_jobs' set-up: random intervals of every motif, with and without the planes supplied."""
    if len(seq) < 8:
        pytest.skip("too short for an interval")
    rs = np.random.RandomState(len(seq) + m_hi)
    with Oracle(seq, m_lo, m_hi) as o:
        o.run_perfect(); o.run_subst(); o.run_anchor_planes()
        planes = {m: o.plane(m) for m in range(m_lo, m_hi + 1)}
    L = len(seq)
    ms = list(range(m_lo, m_hi + 1)) if m_hi - m_lo < 24 else sorted(set(int(x) for x in rs.randint(m_lo, m_hi + 1, 24)) | {m_lo, m_hi})
    seeds = []
    for m in ms:
        for _ in range(12):
            a = int(rs.randint(0, L - 1))
            b = int(min(L, a + 1 + rs.randint(0, min(L, 40 * m + 64))))
            seeds.append((a, b, m, 0))
        seeds.append((0, L, m, 0))
        seeds.append((max(0, L - 3 * m - 5), L, m, 0))
    got = ribbit_amd.host_longest_runs(m_lo, m_hi, seq, np.array(seeds, dtype=ribbit_amd.SEED_DT))
    for (a, b, m, _), g in zip(seeds, got):
        bits = planes[m][a:b].astype(np.int8)
        d = np.diff(np.concatenate(([0], bits, [0])))
        runs = np.flatnonzero(d == -1) - np.flatnonzero(d == 1)
        assert g == (int(runs.max()) if len(runs) else 0), (m, a, b)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))), ids=lambda p: os.path.basename(p)[:-4])
def test_replay_matches_fixture(path):
    g = np.load(path)
    seq = g["seq"].tobytes()
    r = ribbit_amd.host_replay_calls(int(g["m_lo"]), int(g["m_hi"]), seq, g["perfect_calls"], g["subst_calls"])
    assert np.array_equal(r["perfect"].view("<i4"), g["perfect_after_s"].view("<i4"))
    assert np.array_equal(r["subst"].view("<i4"), g["subst_after_s"].view("<i4"))


def test_replay_rejects_short_planes(hip_lib):
    import ctypes as C
    p = ribbit_amd.ScanParams()
    hip_lib.ribbit_scan_params_default(C.byref(p), 2, 100)
    w = np.zeros(4, dtype=np.uint32)
    out = ribbit_amd.SeedLists()
    rc = hip_lib.ribbit_host_replay_calls(C.byref(p), 100, w.ctypes.data, w.ctypes.data, w.ctypes.data, 4, None, 0,
                                          None, 0, None, 0, None, 0, C.byref(out))
    assert rc == -1 and b"planes too short" in hip_lib.ribbit_hip_last_error()
