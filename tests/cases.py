"""Seeded inputs shared by the CPU (oracle) and GPU parity tests."""
import numpy as np

from ribbit_amd.simulate import random_sequence, simulate_sequence


def _rand(n, seed, alphabet=b"ACGT"):
    rs = np.random.RandomState(seed)
    return bytes(np.frombuffer(alphabet, dtype=np.uint8)[rs.randint(0, len(alphabet), size=n)])


def edge_cases():
    """(name, sequence, m_lo, m_hi): the regimes SURVEY.md 3.2 calls out (Q4-Q7) plus tile/word edges."""
    cases = [
        ("empty", b"", 2, 6),
        ("one_base", b"A", 2, 6),
        ("short_7", b"ACGTACG", 2, 6),
        ("all_A_100", b"A" * 100, 2, 10),
        ("all_N_100", b"N" * 100, 2, 10),
        ("poly_AC", b"AC" * 200, 2, 12),
        ("tail_As", _rand(300, 11) + b"A" * 64, 2, 20),          # Q5: trailing A's compare equal to the zero fill
        ("tail_Ns", _rand(300, 12) + b"N" * 40, 2, 20),          # Q5: trailing N's too
        ("lower", (b"acgtt" * 40) + _rand(100, 13).lower(), 2, 8),
        ("n_inside_run", b"CAG" * 30 + b"N" + b"CAG" * 30 + _rand(50, 14), 2, 8),      # Q6/Q7
        ("n_runs", _rand(200, 15) + b"N" * 9 + b"AT" * 40 + b"NN" + b"AT" * 40 + b"N" + _rand(200, 16), 2, 16),
        ("iupac", _rand(500, 17, b"ACGTRYKMN"), 2, 12),
        ("word_edges", _rand(31, 18) + b"ACG" * 11 + _rand(32, 19) + b"TTGCA" * 13 + _rand(1, 20), 2, 9),
        ("exact_32", b"GATTACA" * 4 + b"GATT", 2, 7),
        ("exact_64", b"CT" * 32, 2, 6),
        ("m_3_9", _rand(100, 21) + b"AGGCT" * 20 + _rand(100, 22), 3, 9),
        ("m_5_40", _rand(200, 23) + (_rand(17, 24) * 12) + _rand(200, 25), 5, 40),
        ("m_30_100", _rand(300, 26) + (_rand(64, 27) * 6) + _rand(300, 28) + (_rand(97, 29) * 4) + _rand(100, 30), 30, 100),
        ("long_run_cross_tiles", _rand(100, 31) + b"ACGGT" * 8000 + _rand(100, 32), 2, 12),   # > 2 tiles of 16384
        ("run_to_end", _rand(100, 33) + b"CAT" * 50, 2, 8),
        ("run_from_start", b"GA" * 60 + _rand(100, 34), 2, 8),
    ]
    return cases


def simulated_cases():
    """(name, sequence, m_lo, m_hi) from the seeded generator (BASELINE.json configs, scaled down)."""
    out = []
    seq, _ = simulate_sequence(200_000, 1, 2, 6)
    out.append(("sim_cfg1_200k", seq, 2, 6))
    seq, _ = simulate_sequence(120_000, 2, 2, 100, n_block_rate=0.3, lower_rate=0.2)
    out.append(("sim_cfg2_120k", seq, 2, 100))
    seq, _ = simulate_sequence(60_000, 5, 4, 50, n_block_rate=0.5)
    out.append(("sim_m4_50_60k", seq, 4, 50))
    out.append(("random_n_100k", random_sequence(100_000, 3, n_fraction=0.02, n_run_lo=10, n_run_hi=500), 2, 100))
    return out


def _mutate(unit: bytes, copies: int, rate: float, seed: int) -> bytes:
    """copies x unit with substitutions / 1-base indels at `rate` per base (a degenerate tandem repeat)."""
    rs = np.random.RandomState(seed)
    out = bytearray()
    for c in bytearray(unit * copies):
        u = rs.rand()
        if u < rate * 0.6:
            out.append(b"ACGT"[rs.randint(4)])
        elif u < rate * 0.8:
            continue
        elif u < rate:
            out.append(c); out.append(b"ACGT"[rs.randint(4)])
        else:
            out.append(c)
    return bytes(out)


def large_motif_cases():
    """-M above 110: the anchored kernel needs more than one halo lane per side (BASELINE.json configs[4] runs
    -M 500).  Long units with few copies, degenerate copies (anchors of length up to 2s), all-ones stretches
    longer than a lane (homopolymers), N blocks, and tile edges (> 1 tile at 5 halo lanes = 13824 bases)."""
    big = (_rand(700, 41) + _rand(180, 42) * 4 + _rand(300, 43) + _mutate(_rand(140, 44), 6, 0.04, 45) + _rand(500, 46)
           + b"A" * 1300 + _rand(200, 47) + _mutate(_rand(37, 48), 40, 0.06, 49) + b"N" * 300 + _rand(64, 50) * 30
           + _rand(900, 51) + _mutate(_rand(411, 52), 4, 0.02, 53) + _rand(400, 54))
    long = _rand(9000, 55) + _mutate(_rand(260, 56), 9, 0.05, 57) + _rand(3000, 58) + b"CA" * 900 + _rand(12000, 59) \
        + _mutate(_rand(95, 60), 30, 0.08, 61) + _rand(2500, 62)
    return [
        ("M200_mixed", big, 2, 200),
        ("M500_mixed", big, 100, 500),
        ("M500_two_tiles", long, 60, 500),
        ("M990_long_units", _rand(1500, 63) + _rand(800, 64) * 3 + _rand(600, 65) + _mutate(_rand(600, 66), 4, 0.03, 67) + _rand(900, 68), 500, 990),
        ("M111_first_above_one_halo_lane", big[:6000], 90, 111),
    ]


def structured_cases():
    """Long regular structures: runs that span many kernel tiles (the pairing kernel's search for a run's END over
    empty tiles, all-ones lanes in the anchored kernel's run-length scan), periodic N, a record that is mostly N."""
    homo = _rand(700, 71) + b"A" * 40_000 + _rand(900, 72)        # > 2 kernel tiles; the merges are quadratic in such a run
    dinuc = bytearray(b"AC" * 30_000)
    for p in range(4_999, len(dinuc), 5_000):
        dinuc[p] = ord("G")                                   # one mismatch every 5 kb
    dinuc = _rand(500, 73) + bytes(dinuc) + _rand(500, 74)
    n_every = bytearray(simulate_sequence(150_000, 75, 2, 40)[0])
    n_every[::1000] = b"N" * len(n_every[::1000])
    mostly_n = b"N" * 65_000 + b"CAG" * 400 + b"N" * 70_000 + _rand(3_000, 76) + b"TTAGGG" * 300 + b"N" * 20_000
    return [
        ("homopolymer_40k_small_m", homo, 2, 20),
        ("homopolymer_40k_large_m", homo, 100, 300),
        ("dinucleotide_60k_sparse_mismatches", dinuc, 2, 12),
        ("n_every_1000", bytes(n_every), 2, 40),
        ("mostly_n", mostly_n, 2, 30),
    ]
