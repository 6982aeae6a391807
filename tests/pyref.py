"""Second, independent restatement (pure Python/numpy, small inputs only) of the *scan* loops of
the reference, used to cross-check the C oracle's transcription of them.  Test-only.

Covers: encode + sweep (fasta_utils.cpp:78-122), the perfect run scan
(parse_perfect_shiftxor.cpp:146-226), the window FSM (parse_substitute_shiftxor.cpp:391-577,
parse_anchored_shiftxor.cpp:538-726) and the anchor planes (parse_anchored_shiftxor.cpp:20-56,
fasta_utils.cpp:143-161).  The addSeed* merges are NOT restated here.
"""
import numpy as np

_CODE = np.full(256, 255, dtype=np.uint8)
for ch, v in ((b"Aa", 0), (b"Cc", 1), (b"Gg", 2), (b"Tt", 3)):
    for c in ch:
        _CODE[c] = v


def encode(seq: bytes):
    raw = _CODE[np.frombuffer(seq, dtype=np.uint8)] if len(seq) else np.zeros(0, np.uint8)
    nmask = (raw == 255).astype(np.uint8)
    code = np.where(raw == 255, 0, raw).astype(np.uint8)
    return code, nmask


def plane(code: np.ndarray, s: int):
    L = len(code)
    partner = np.zeros(L, dtype=np.uint8)
    if s < L:
        partner[: L - s] = code[s:]
    return (code == partner).astype(np.uint8)


def shift_range(m_lo, m_hi):
    return (m_lo - 2 if m_lo > 2 else 1), m_hi + 2


def runs_of_ones(y: np.ndarray):
    """maximal runs [s, e) of ones in y"""
    if len(y) == 0:
        return []
    d = np.diff(np.concatenate(([0], y.astype(np.int8), [0])))
    return list(zip(np.flatnonzero(d == 1).tolist(), np.flatnonzero(d == -1).tolist()))


def perfect_calls(seq: bytes, m_lo: int, m_hi: int):
    """(pos, mlen, start, end) in the order processShiftXORsPerfect makes its addSeed calls."""
    code, nmask = encode(seq)
    L = len(code)
    min_shift, _ = shift_range(m_lo, m_hi)
    inloop, flush = [], []
    for m in range(m_lo, m_hi + 1):
        c1 = 12 - m if m <= 6 else m
        c2 = 12 - m if m <= 6 else m + (m - min_shift)
        y = plane(code, m) & (1 - nmask)
        for s, e in runs_of_ones(y):
            if e == L:
                if (L - 1) - s >= c1:
                    flush.append((L, m, s, L - 1))
            elif nmask[e]:
                if e - s >= c2:
                    inloop.append((e, m, s, e))
            elif e - s >= c1:
                inloop.append((e, m, s, e))
    inloop.sort(key=lambda t: (t[0], t[1]))
    flush.sort(key=lambda t: t[1])
    return inloop + flush


def window_calls(planes: dict, nmask: np.ndarray, m_lo: int, m_hi: int, threshold: int, wl: int = 8):
    """Literal per-position transcription of the window FSM; returns the addSeed call log."""
    L = len(nmask)
    nm = m_hi - m_lo + 1
    pend_s = [-1] * nm
    pend_e = [-1] * nm
    cur = [-1] * nm
    win = [0] * nm
    mask = (1 << wl) - 1
    calls = []
    valid = 0
    wpos = -wl
    for p in range(L):
        wpos += 1
        if nmask[p]:
            for d in range(nm):
                if cur[d] != -1:
                    cur[d] = wpos
                    if pend_e[d] != -1 and pend_e[d] < cur[d]:
                        calls.append((p, m_lo + d, pend_s[d], pend_e[d]))
                        pend_s[d] = pend_e[d] = -1
                win[d] = 0
                cur[d] = -1
            valid = 0
            continue
        valid += 1
        for d in range(nm):
            win[d] = ((win[d] << 1) | int(planes[m_lo + d][p])) & mask
        if valid < wl:
            continue
        for d in range(nm):
            if bin(win[d]).count("1") >= threshold:
                if cur[d] == -1:
                    cur[d] = wpos
                    if pend_e[d] != -1 and pend_e[d] < cur[d]:
                        calls.append((p, m_lo + d, pend_s[d], pend_e[d]))
                        pend_s[d] = pend_e[d] = -1
            elif cur[d] != -1:
                if pend_s[d] == -1:
                    pend_s[d] = cur[d]
                pend_e[d] = wpos + wl - 1
                cur[d] = -1
            elif pend_e[d] != -1 and pend_e[d] < wpos:
                calls.append((p, m_lo + d, pend_s[d], pend_e[d]))
                pend_s[d] = pend_e[d] = -1
    for d in range(nm):
        m = m_lo + d
        if pend_e[d] == -1:
            if cur[d] != -1:
                calls.append((L, m, cur[d], L))
        elif cur[d] == -1:
            calls.append((L, m, pend_s[d], pend_e[d]))
        elif pend_e[d] >= cur[d] - m:
            calls.append((L, m, pend_s[d], L))
        else:
            calls.append((L, m, pend_s[d], pend_e[d]))
            calls.append((L, m, cur[d], L))
    return calls


def anchor_plane(x: np.ndarray, s: int, anchor_size: int = 3):
    """generateAnchoredShiftXORs for one shift: runs with 3 <= len < 2s closed by a zero at p <= L-1-s."""
    L = len(x)
    a = np.zeros(L, dtype=np.uint8)
    start = -1
    for p in range(0, L - s):
        if x[p]:
            if start == -1:
                start = p
        else:
            if start != -1 and anchor_size <= p - start < 2 * s:
                a[start:p] = 1
            start = -1
    return a


def anchored_planes(seq: bytes, m_lo: int, m_hi: int):
    code, _ = encode(seq)
    lo, hi = shift_range(m_lo, m_hi)
    X = {s: plane(code, s) for s in range(lo, hi + 1)}
    A = {s: anchor_plane(X[s], s) for s in range(lo, hi + 1)}
    XA = dict(X)
    for m in range(m_lo, m_hi + 1):
        acc = X[m].copy()
        for i in range(m - 2 if m > 2 else 1, m + 3):
            if i != m:
                acc |= A[i]
        XA[m] = acc
    return X, A, XA
