"""Seeded synthetic FASTA generator for the ribbit scan benchmarks and tests.

Restates the *distribution* of the reference's data_simulation/simulate_data.py (which is
unseeded and depends on two tables that are not shipped):

* loci are separated by buffers of U[500,3000] bases cut from the start of a fixed 547-base
  string repeated cyclically (simulate_data.py:11-17,121-125) -- here the 547 bases are drawn
  once from a fixed LCG instead of copying the reference's string;
* each locus picks a motif size from a proportions table (stand-in below, the reference's
  proportions.tsv is absent), units from choose_num_units (simulate_data.py:20-24), a partial
  suffix int(U{0..9}/10 * m) (:129) and a motif of that size (stand-in for the absent
  HG38 motif list: a uniformly random primitive word);
* impurity U[5,15] % of the repeat length (:113-114,136-137) capped at one mutation per motif
  unit because int(1-0.75)*m == 0 (:139-140); mutation kinds S/I/D = 80/10/10 (:10);
* 80-column FASTA lines (:172-174).

Extras the reference simulator does not have (used by the parity tests): N blocks, lower-case
stretches, and a uniform-random control sequence.
"""
from __future__ import annotations

import numpy as np

_ALPHABET = b"ACGT"


def _fixed_buffer_unit(n: int = 547) -> bytes:
    # fixed 547-base unit from a 64-bit LCG (Knuth MMIX constants); deterministic everywhere
    x = 0x9E3779B97F4A7C15
    out = bytearray()
    for _ in range(n):
        x = (x * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        out.append(_ALPHABET[(x >> 33) & 3])
    return bytes(out)


BUFFER_UNIT = _fixed_buffer_unit()


def motif_size_weights(m_lo: int, m_hi: int) -> tuple[np.ndarray, np.ndarray]:
    """Stand-in for proportions.tsv: short motifs dominate, long ones form a thin tail."""
    sizes = np.arange(m_lo, m_hi + 1)
    w = np.empty(len(sizes), dtype=np.float64)
    for i, m in enumerate(sizes):
        if m == 2:
            w[i] = 25.0
        elif m == 3:
            w[i] = 10.0
        elif m == 4:
            w[i] = 20.0
        elif m == 5:
            w[i] = 10.0
        elif m == 6:
            w[i] = 8.0
        elif m <= 10:
            w[i] = 2.0
        elif m <= 30:
            w[i] = 0.5
        else:
            w[i] = 0.1
    return sizes, w / w.sum()


def _choose_num_units(rs: np.random.RandomState, m: int) -> int:
    if m == 2:
        return int(rs.randint(6, 101))
    if m == 3:
        return int(rs.randint(4, 101))
    if m <= 50:
        return int(rs.randint(3, 101))
    return int(rs.randint(2, 11))


def _primitive_motif(rs: np.random.RandomState, m: int) -> bytes:
    while True:
        word = bytes(_ALPHABET[i] for i in rs.randint(0, 4, size=m))
        if not any(m % d == 0 and word == word[:d] * (m // d) for d in range(1, m)):
            return word


def _mutate(rs: np.random.RandomState, repeat: bytes, m: int, units: int,
            min_purity: float, max_purity: float) -> bytes:
    rlen = len(repeat)
    lo = int(100 * (1 - max_purity))
    hi = int(100 * (1 - min_purity))
    impurity = int(rs.randint(lo, hi + 1))
    n_mut = min(int((impurity / 100) * rlen), units)   # one mutation per motif unit at most
    if rlen < 2:
        return repeat
    chosen: dict[int, int] = {}
    used_units: set[int] = set()
    guard = 0
    while len(chosen) < n_mut and guard < 100 * (n_mut + 1):
        guard += 1
        pos = int(rs.randint(1, rlen))
        if pos in chosen or pos // m in used_units:
            continue
        r = int(rs.randint(0, 100))
        chosen[pos] = 0 if r < 80 else (1 if r < 90 else 2)   # 0 S, 1 I, 2 D
        used_units.add(pos // m)
    out = bytearray()
    x = 0
    for pos in sorted(chosen):
        out += repeat[x:pos]
        kind = chosen[pos]
        if kind == 2:
            x = pos + 1
        elif kind == 0:
            alts = [b for b in _ALPHABET if b != repeat[pos]]
            out.append(alts[int(rs.randint(0, 3))])
            x = pos + 1
        else:
            out.append(_ALPHABET[int(rs.randint(0, 4))])
            x = pos
    out += repeat[x:]
    return bytes(out)


def simulate_sequence(total_len: int, seed: int, m_lo: int = 2, m_hi: int = 100,
                      min_purity: float = 0.85, max_purity: float = 0.95,
                      n_block_rate: float = 0.0, lower_rate: float = 0.0):
    """Return (sequence bytes of exactly total_len, truth list[(start, end, motif size, motif)]).

    n_block_rate: expected N blocks per locus (block length U[1,200]); lower_rate: probability a
    locus is written in lower case.
    """
    rs = np.random.RandomState(seed)
    sizes, weights = motif_size_weights(m_lo, m_hi)
    parts: list[bytes] = []
    truth = []
    pos = 0
    while pos < total_len:
        blen = int(rs.randint(500, 3001))
        buf = (BUFFER_UNIT * (blen // len(BUFFER_UNIT) + 1))[:blen]
        parts.append(buf)
        pos += blen
        if n_block_rate > 0 and rs.random_sample() < n_block_rate:
            nlen = int(rs.randint(1, 201))
            parts.append(b"N" * nlen)
            pos += nlen
        m = int(rs.choice(sizes, p=weights))
        units = _choose_num_units(rs, m)
        suffix = int((int(rs.randint(0, 10)) / 10) * m)
        rlen = m * units + suffix
        if suffix > 0.75 * m:
            units += 1
        motif = _primitive_motif(rs, m)
        repeat = (motif * (units + 1))[:rlen]
        mutated = _mutate(rs, repeat, m, units, min_purity, max_purity)
        if lower_rate > 0 and rs.random_sample() < lower_rate:
            mutated = mutated.lower()
        truth.append((pos, pos + len(mutated), m, motif.decode()))
        parts.append(mutated)
        pos += len(mutated)
    seq = b"".join(parts)[:total_len]
    truth = [t for t in truth if t[1] <= total_len]
    return seq, truth


def random_sequence(total_len: int, seed: int, n_fraction: float = 0.001,
                    n_run_lo: int = 1000, n_run_hi: int = 10000) -> bytes:
    """Uniform ACGT control with a fraction of the bases inside N runs."""
    rs = np.random.RandomState(seed)
    arr = np.frombuffer(_ALPHABET, dtype=np.uint8)[rs.randint(0, 4, size=total_len, dtype=np.uint8)].copy()
    target = int(total_len * n_fraction)
    placed = 0
    while placed < target and total_len > n_run_lo:
        rl = int(rs.randint(n_run_lo, n_run_hi + 1))
        st = int(rs.randint(0, max(1, total_len - rl)))
        arr[st:st + rl] = ord("N")
        placed += rl
    return arr.tobytes()


def write_fasta(path: str, records: list[tuple[str, bytes]], width: int = 80) -> None:
    with open(path, "wb") as fh:
        for name, seq in records:
            fh.write(b">" + name.encode() + b"\n")
            for i in range(0, len(seq), width):
                fh.write(seq[i:i + width] + b"\n")


# BASELINE.json configs[2] / configs[3] stand-in (GRCh38 itself is on no box): 24 records with the primary
# assembly's lengths.  One definition for everything that makes or checks these records -- bench.py's chr1_full_path leg,
# tools/grch38_shape.py, tests/golden/make_full_size_digests.py and the digests under tests/golden/ -- so that they cannot drift apart.
GRCH38_RECORDS = [("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555), ("chr5", 181538259), ("chr6", 170805979),
                  ("chr7", 159345973), ("chr8", 145138636), ("chr9", 138394717), ("chr10", 133797422), ("chr11", 135086622), ("chr12", 133275309),
                  ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189), ("chr16", 90338345), ("chr17", 83257441), ("chr18", 80373285),
                  ("chr19", 58617616), ("chr20", 64444167), ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415)]


def grch38_shaped_record(k: int, bases: int | None = None, m_lo: int = 2, m_hi: int = 100) -> bytes:
    """Record k of the GRCh38-shaped set (k = 0: the chromosome-1-sized record of bench.py): generator seed 4 + 1000 k, blocks
    of N at both ends and one 3-Mbp centromere-like block of N in the middle.  `bases` overrides the length (scaled sets)."""
    if bases is None:
        bases = GRCH38_RECORDS[k][1]
    seq, _ = simulate_sequence(bases, 4 + 1000 * k, m_lo, m_hi)
    b = np.frombuffer(seq, dtype=np.uint8).copy()
    edge = min(10_000, bases // 100)
    b[:edge] = ord("N")
    b[bases - edge:] = ord("N")
    cen = min(3_000_000, bases // 50)
    b[bases // 2:bases // 2 + cen] = ord("N")
    return b.tobytes()


def m500_record(bases: int) -> bytes:
    """BASELINE.json configs[4]'s motif range on one record: generator motifs 2..500, blocks of N and lower-case loci (seed 77).  One
    definition for tests/test_m500_gpu.py, bench.py's `m500_full_path` leg and the digest under tests/golden/."""
    seq, _ = simulate_sequence(bases, 77, 2, 500, n_block_rate=0.1, lower_rate=0.1)
    return seq
