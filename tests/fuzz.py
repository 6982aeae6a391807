"""Seeded generator of small adversarial records for the fuzz tests (CPU and GPU): low-complexity DNA in which
the seed merges are busiest -- nested / overlapping repeats of related motif sizes, degenerate copies, N blocks,
homopolymers, lower case, records shorter than a window, motif ranges anywhere in 1..990."""
import numpy as np

ALPHABETS = [b"ACGT", b"ACGT", b"AC", b"ACG", b"AT", b"ACGTN", b"acgtACGT"]


def _piece(rs):
    kind = rs.randint(0, 9)
    alpha = ALPHABETS[rs.randint(len(ALPHABETS))]
    pick = lambda n: bytes(np.frombuffer(alpha, dtype=np.uint8)[rs.randint(0, len(alpha), size=n)])
    if kind <= 1:                                   # random background
        return pick(rs.randint(1, 400))
    if kind == 2:                                   # homopolymer / N block
        return bytes([b"ACGTN"[rs.randint(5)]]) * rs.randint(1, 300)
    unit = pick(int(rs.choice([1, 2, 3, 4, 5, 6, 7, 9, 12, 15, 24, 31, 32, 33, 50, 64, 97, 130])))
    copies = rs.randint(2, max(3, 400 // len(unit)))
    body = bytearray(unit * copies)
    if kind >= 5:                                   # degenerate copies: substitutions and 1-base indels
        rate = rs.choice([0.01, 0.03, 0.08, 0.15])
        out = bytearray()
        for c in body:
            u = rs.rand()
            if u < rate * 0.6:
                out.append(b"ACGT"[rs.randint(4)])
            elif u < rate * 0.8:
                continue
            elif u < rate:
                out.append(c); out.append(b"ACGT"[rs.randint(4)])
            else:
                out.append(c)
        body = out
    if kind == 8:                                   # a related motif right next to it (unit doubled / halved / rotated)
        rel = (unit * 2)[1:len(unit) + 1] if rs.rand() < 0.5 else unit[:max(1, len(unit) // 2)]
        body += rel * rs.randint(2, 12)
    return bytes(body)


def fuzz_case(seed: int, scale: int = 1):
    """-> (sequence, m_lo, m_hi); scale > 1 stretches the records over many kernel tiles"""
    rs = np.random.RandomState(seed)
    target = scale * int(rs.choice([0, 1, 5, 9, 40, 300, 2000, 6000, 20000], p=[.01, .01, .02, .02, .04, .2, .4, .2, .1]))
    parts, n = [], 0
    while n < target:
        p = _piece(rs)
        parts.append(p)
        n += len(p)
    seq = b"".join(parts)[:target]
    style = rs.randint(0, 10)
    if style <= 5:
        m_lo = int(rs.randint(1, 8)); m_hi = int(rs.randint(m_lo, 40))
    elif style <= 7:
        m_lo = int(rs.randint(2, 30)); m_hi = int(rs.randint(m_lo, 140))
    elif style == 8:
        m_lo = int(rs.randint(2, 120)); m_hi = int(min(990, m_lo + rs.randint(0, 60)))
    else:
        m_lo = int(rs.randint(100, 400)); m_hi = int(min(990, m_lo + rs.randint(0, 300)))
    return seq, m_lo, m_hi
