"""Test-side numpy restatement of what ribbit_amd/csrc/window_stage.hip computes on the GPU: the window state machine
of parse_substitute_shiftxor.cpp:430-574 / parse_anchored_shiftxor.cpp:580-723 as a LOCAL rule over pass-streaks
(which streak joins its predecessor's group, one call per group, where it is made) plus the cursor bounds of the
compact form.  Checked against the oracle's sequential state machine on CPU (tests/test_streaks.py); the GPU tests
then check the kernels against the oracle directly.  Test-only."""
import numpy as np

ZERO, N, EOS = 0, 1, 2


def eval_mask(nmask):
    """E[q] = window [q, q+7] lies inside the record and holds no N"""
    L = len(nmask)
    e = np.zeros(L + 16, bool)
    nq = max(L - 7, 0)
    if nq:
        cs = np.concatenate(([0], np.cumsum(nmask.astype(np.int64))))
        e[:nq] = (cs[8:8 + nq] - cs[:nq]) == 0
    return e


def first_evaluated_table(e):
    """t[x] = smallest q >= x with e[q], -1 if none"""
    n = len(e)
    t = np.full(n + 1, -1, np.int64)
    for q in range(n - 1, -1, -1):
        t[q] = q if e[q] else t[q + 1]
    return t


def streaks_of(events, counts, m_lo):
    """[(mlen, start, end, term)] motif-major from packed START / END events (tests/pyevents.py layout)"""
    out = []
    off = 0
    for mi, c in enumerate(counts):
        seg = events[off:off + int(c)]
        off += int(c)
        pos = (seg & np.uint64(0xFFFFFFFF)).astype(np.int64)
        kind = ((seg >> np.uint64(48)) & np.uint64(0xF)).astype(np.int64)
        assert len(seg) % 2 == 0 and np.all(kind[0::2] == 0) and np.all(kind[1::2] != 0)
        for s, e, k in zip(pos[0::2], pos[1::2], kind[1::2]):
            out.append((m_lo + mi, int(s), int(e), int(k) - 1))
    return out


def calls_from_streaks(streaks, nmask, min_span=None):
    """-> (in-loop calls sorted by (pos, mlen), flush calls by motif, per in-loop call: edge flag)
    every call is (pos, mlen, start, end)"""
    L = len(nmask)
    e = eval_mask(nmask)
    first = first_evaluated_table(e)
    n = len(streaks)

    def joins(i):   # streak i joins the group of streak i-1
        if i == 0:
            return False
        pm, ps, pe, pt = streaks[i - 1]
        m, s, _, _ = streaks[i]
        return pm == m and pt == ZERO and pe + 7 >= s

    group_start = [0] * n
    for i in range(n):
        group_start[i] = group_start[i - 1] if joins(i) else streaks[i][1]
    inloop, flush = [], []
    for i in range(n):
        m, s, en, term = streaks[i]
        if term == ZERO and i + 1 < n and joins(i + 1):
            continue
        if term == ZERO:
            start, end = group_start[i], en + 7
            q = first[min(end + 1, len(first) - 1)]
            if q < 0:
                flush.append((L, m, start, end))
            else:
                inloop.append((q + 7, m, start, end))
        elif term == N:
            if joins(i):
                start, end = group_start[i], streaks[i - 1][2] + 7
                if end < en:
                    inloop.append((en + 7, m, start, end))
                else:
                    q = first[min(end + 1, len(first) - 1)]
                    if q < 0:
                        flush.append((L, m, start, end))
                    else:
                        inloop.append((q + 7, m, start, end))
        else:
            flush.append((L, m, group_start[i] if joins(i) else s, L))
    inloop.sort(key=lambda c: (c[0], c[1]))
    flush.sort(key=lambda c: c[1])
    edge = [bool((pos < L and nmask[pos]) or pos < 8 or not e[pos - 8]) for pos, _, _, _ in inloop]
    return inloop, flush, edge


def compact_bounds(inloop, edge, min_span):
    """the compact form window_stage.hip hands the merges: kept calls in order, each with its cursor bound
    (-1 where no earlier call can reach beyond the call's own end)"""
    kept, bounds = [], []
    last_ordinary = -1          # last position at which an ordinary (non-edge) call was made
    edge_max_end = -1           # largest end among the edge calls so far
    i = 0
    n = len(inloop)
    while i < n:
        j = i
        while j < n and inloop[j][0] == inloop[i][0]:
            j += 1
        pos = inloop[i][0]
        for k in range(i, j):
            _, m, s, en = inloop[k]
            if edge[k]:
                b = max(edge_max_end, last_ordinary - 8 if last_ordinary >= 8 else -1)
                if en - s >= min_span(m):
                    kept.append(inloop[k]); bounds.append(b)
                edge_max_end = max(edge_max_end, en)
            else:
                assert en == pos - 8, "an ordinary call ends 8 before the position it is made at"
                if en - s >= min_span(m):
                    kept.append(inloop[k]); bounds.append(-1)
        if not all(edge[i:j]):
            assert not any(edge[i:j]), "edge is a property of the position"
            last_ordinary = pos
        i = j
    return kept, bounds
