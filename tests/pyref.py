"""Second, independent restatement (pure Python/numpy, small inputs only) of the *scan* loops of
the reference and of the anchored stage's list merge, used to cross-check the C oracle's transcription of them.  Test-only.

Covers: encode + sweep (fasta_utils.cpp:78-122), the perfect run scan
(parse_perfect_shiftxor.cpp:146-226), the window FSM (parse_substitute_shiftxor.cpp:391-577,
parse_anchored_shiftxor.cpp:538-726) and the anchor planes (parse_anchored_shiftxor.cpp:20-56,
fasta_utils.cpp:143-161), and -- at the end of the file -- addSeedToSeedPositionsAnchored + mergeAllLists
(parse_anchored_shiftxor.cpp:113-534, merge_types.cpp:11-189).  The perfect and substitution stages' merges are NOT
restated here.
"""
import collections

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


# ---------------------------------------------------------------------------------------------------------------------
# The anchored stage's list merge: addSeedToSeedPositionsAnchored (parse_anchored_shiftxor.cpp:113-534) with its
# candidate walk mergeAllLists (merge_types.cpp:11-189), restated a second time, independently of oracle/ribbit_oracle.c
# and of ribbit_amd/csrc/seed_lists.cpp, from the reference text.  Seeds are [start, end, motif, type] lists, edited in
# place as the reference edits its tuples.  Where the reference's behaviour is undefined the two guards of the oracle
# are taken over and counted (D1: an empty substitution list counts as exhausted; D2: a by-counter index beyond a list
# leaves the values of the previous step / skips the write), so the fixtures' `guard_hits` can be compared too.
RANK_P, RANK_Q, RANK_S, RANK_F, RANK_C, RANK_A, RANK_N = 5, 4, 3, 2, 1, 0, -1       # global_variables.cpp:29-35
_U32 = 0xFFFFFFFF


def _i32(x):
    x &= _U32
    return x - (1 << 32) if x >> 31 else x


def anchored_cutoff(m):
    """seedlen_cutoffs of processShiftXORsAnchored (parse_anchored_shiftxor.cpp:572-573)"""
    c = m if m > 6 else 10
    if m >= 10:
        c = int(0.9 * m)
    return c


class AnchoredMerge:
    def __init__(self, perfect, subst, planes, length):
        self.P = [list(map(int, s)) for s in perfect]
        self.S = [list(map(int, s)) for s in subst]
        self.A = []
        self.planes = planes        # shift -> one byte per position (composed planes for the motif sizes in range)
        self.L = int(length)
        self.guards = 0
        self.met = collections.Counter()      # how often the paths the tests want to have seen were taken

    # ---- merge_types.cpp:11-189
    def candidates(self, from_p, from_s, seed_start):
        P, S, A = self.P, self.S, self.A
        sp = []                                   # (rank, index): perfect and substitution seeds, by end, descending
        p_done = len(P) == 0                      # :24
        s_done = False
        if len(S) == 0:                           # D1
            s_done = True
            self.guards += 1
        pi, si = from_p, from_s

        def walk_down(lst, idx, rank, out):
            """the single-list loops (:30-45, :47-62, :108-121, :143-158): entries ending at or after seed_start,
            retired ones left out, until the list's beginning or the first entry that ends before seed_start"""
            while True:
                end, typ = lst[idx][1], lst[idx][3]
                if end >= seed_start:
                    if typ != RANK_N:
                        out.append((rank, idx))
                    idx -= 1
                if idx < 0 or end < seed_start:
                    return idx

        while not (p_done and s_done):            # :28
            if s_done:
                pi = walk_down(P, pi, RANK_P, sp)
                p_done = True
            elif p_done:
                si = walk_down(S, si, RANK_S, sp)
                s_done = True
            else:                                 # :64-94: both lists open -- the larger end first, perfect on a tie
                p_end, s_end = P[pi][1], S[si][1]
                if s_end > p_end:
                    if S[si][3] != RANK_N:
                        sp.append((RANK_S, si))
                    si -= 1
                else:
                    if P[pi][3] != RANK_N:
                        sp.append((RANK_P, pi))
                    pi -= 1
                if pi < 0 or p_end < seed_start:
                    p_done = True
                if si < 0 or s_end < seed_start:
                    s_done = True

        out = []
        if len(A) == 0:                           # :103
            return list(sp)
        if not sp:                                # :107
            walk_down(A, len(A) - 1, RANK_A, out)
            return out

        def sp_end(k):
            rank, idx = sp[k]
            return (P if rank == RANK_P else S)[idx][1]

        k, ai = len(sp) - 1, len(A) - 1           # :99: the perfect/substitution candidates are taken from their LAST
        k_done = a_done = False
        while not (k_done and a_done):            # :124
            if a_done:                            # :125-141 (retired entries were left out when sp was made)
                while True:
                    e = sp_end(k)
                    if e >= seed_start:
                        out.append(sp[k])
                        k -= 1
                    if k < 0 or e < seed_start:
                        k_done = True
                        break
            elif k_done:
                walk_down(A, ai, RANK_A, out)
                a_done = True
            else:                                 # :160-186: note that retired anchored seeds ARE listed here
                e, a_end = sp_end(k), A[ai][1]
                if a_end > e:
                    out.append((RANK_A, ai))
                    ai -= 1
                else:
                    out.append(sp[k])
                    k -= 1
                if k < 0 or e < seed_start:
                    k_done = True
                if ai < 0 or a_end < seed_start:
                    a_done = True
        return out

    # ---- parse_anchored_shiftxor.cpp:59-71
    def retain_nested(self, start, end, nested_m, parent_m):
        nested = int(self.planes[nested_m][start:end].sum())
        parent = int(self.planes[parent_m][start:end].sum())
        return nested >= parent

    def _retire_by_type(self, last_type, i):
        """the `if (last_type == RANK_P) ... else if (last_type == RANK_S || last_type == RANK_Q)` pairs"""
        if last_type == RANK_P:
            self.P[i][3] = RANK_N
        elif last_type in (RANK_S, RANK_Q):
            self.S[i][3] = RANK_N

    # ---- parse_anchored_shiftxor.cpp:113-534
    def add(self, seed_start, seed_end, m, frm, seed_type, depth=0):
        P, S, A = self.P, self.S, self.A
        fp, fs = frm
        if depth:
            self.met["added again by a nested call"] += 1
        # :132-152 the cursors: forward while the entry starts at or before seed_end, never beyond the last entry
        for lst, which in ((P, 0), (S, 1)):
            f = fp if which == 0 else fs
            for i in range(f, len(lst)):
                if lst[i][0] > seed_end or f == len(lst) - 1:
                    break
                f += 1
            if which == 0:
                fp = f
            else:
                fs = f
        advanced = (fp, fs)
        if seed_end - seed_start < anchored_cutoff(m):        # :153
            return advanced

        cand = self.candidates(fp, fs, seed_start)            # :156
        seed_rend = seed_end + m
        seed_len = seed_end - seed_start
        seed_rlen = seed_len + m
        nonfactor_types, factor_types, factor_sizes = [], [], []
        last_start = last_end = last_rend = last_mlen = None  # the reference's are uninitialised

        for rank, i in cand:
            src = P if rank == RANK_P else S if rank == RANK_S else A
            last_start, last_end, last_mlen, last_type = src[i]
            last_rend = last_end + last_mlen
            if last_end < seed_start:                         # :203
                break
            if last_type == RANK_N:
                continue
            if seed_end < last_start:
                continue
            last_len = last_end - last_start
            last_rlen = last_rend - last_start

            if seed_start == last_start and seed_end == last_end:                       # :215 same interval
                if seed_type == RANK_A and last_type > RANK_A:
                    return advanced
                if seed_type == RANK_C and last_type == RANK_A:
                    A[i][3] = RANK_N
                continue

            if last_start <= seed_start and seed_end <= last_end:                       # :231 inside an older seed
                if last_type > seed_type:
                    return advanced
                if seed_type == RANK_C and last_type == RANK_A:
                    continue
                if (seed_type, last_type) in ((RANK_A, RANK_A), (RANK_C, RANK_C)):
                    if m % last_mlen == 0 and m != 4:
                        return advanced
                    if last_mlen % m == 0 and last_mlen != 4:
                        if seed_rlen >= last_mlen - 1 or seed_rlen >= last_len:
                            A[i][3] = RANK_N
                            return self.add(last_start, last_end, m, frm, seed_type, depth + 1)
                        continue
                    if not self.retain_nested(seed_start, seed_end, m, last_mlen):
                        return advanced
                continue

            if seed_start <= last_start and last_end <= seed_end:                       # :266 around an older seed
                if last_type > seed_type:
                    if m % last_mlen == 0:
                        if last_rlen >= m - 2 or last_rlen >= seed_len - 2:
                            self._retire_by_type(last_type, i)
                            return self.add(seed_start, seed_end, last_mlen, frm, RANK_C, depth + 1)
                        factor_types.append(last_type)
                        factor_sizes.append(last_mlen)
                    elif last_mlen % m == 0:
                        if last_mlen >= 4 * m or last_len >= 4 * m:
                            self._retire_by_type(last_type, i)
                            return self.add(seed_start, seed_end, m, frm, RANK_C, depth + 1)
                        # (parentof_subperf_multiple: collected, never read)
                    elif last_mlen > m:
                        if last_mlen >= 4 * m or last_len >= 4 * m:
                            self._retire_by_type(last_type, i)
                            return self.add(seed_start, seed_end, m, frm, RANK_C, depth + 1)
                    else:
                        nonfactor_types.append(last_type)
                elif seed_type == RANK_C and last_type == RANK_A:
                    A[i][3] = RANK_N
                elif (seed_type, last_type) in ((RANK_A, RANK_A), (RANK_C, RANK_C)):
                    if last_mlen == m:
                        A[i][3] = RANK_N
                    elif not self.retain_nested(last_start, last_end, last_mlen, m):
                        A[i][3] = RANK_N
                    elif m % last_mlen == 0:
                        if last_rlen >= m - 2 or last_rlen >= seed_len - 2:
                            A[i][3] = RANK_N
                            return self.add(seed_start, seed_end, last_mlen, frm, seed_type, depth + 1)
                continue

            # :354 partial overlap
            if last_start < seed_start:
                reach = last_rend if last_mlen <= m else last_end
                overlap = (seed_end if seed_end <= reach else reach) - seed_start
                merge_start, merge_end = last_start, seed_end
            else:
                reach = seed_rend if m <= last_mlen else seed_end
                overlap = (last_end if last_end <= reach else reach) - last_start
                merge_start, merge_end = seed_start, last_end

            if seed_type == RANK_A and last_type > RANK_C:                              # :379
                if m == last_mlen and overlap >= 4 * m:
                    self._retire_by_type(last_type, i)
                    return self.add(merge_start, merge_end, m, frm, RANK_C, depth + 1)
                if not (m % last_mlen == 0 or last_mlen % m == 0):
                    if overlap >= m - 1 or overlap >= seed_len - 1:
                        return advanced
            elif seed_type in (RANK_A, RANK_C) and last_type in (RANK_A, RANK_C):       # :401
                if m == last_mlen:
                    # (`seed_type == ... ? RANK_C : RANK_A;` at :406 ff. compares and discards: the type is unchanged)
                    if last_len >= seed_len:
                        joins = ((seed_len >= 3 * m and (overlap >= 3 * m - 1 or overlap >= seed_len - 1)) or
                                 (seed_len < 3 * m and (overlap >= m - 1 or overlap >= seed_len - 1)))
                    else:
                        joins = ((last_len >= 3 * last_mlen and (overlap >= 3 * last_mlen - 1 or overlap >= last_len - 1)) or
                                 (seed_len < 3 * last_mlen and (overlap >= last_mlen - 1 or overlap >= last_len - 1)))
                    if joins:
                        A[i][3] = RANK_N
                        return self.add(merge_start, merge_end, last_mlen, frm, seed_type, depth + 1)

        # :438-468 coverage by the perfect / substitution seeds inside that are neither factor nor multiple.  The lists
        # are read at the LOOP COUNTER j, not at the seed's index; a merged-type (RANK_Q) seed reads nothing and the
        # values of the step before stay.  prev_start is a uint32_t and the comparisons and differences with it are unsigned.
        def head(j, ktype):
            nonlocal last_start, last_end, last_rend, last_mlen
            lst = P if ktype == RANK_P else S if ktype == RANK_S else None
            if lst is None:
                self.met["merged-type seed: values of the step before"] += 1
                return
            if j >= len(lst):                     # D2
                self.guards += 1
                return
            last_start, last_end, last_mlen = lst[j][0], lst[j][1], lst[j][2]
            last_rend = last_end + last_mlen

        def covered(total, prev_start):
            if (last_rend & _U32) >= prev_start:
                return _i32(total + ((prev_start - last_start) & _U32))
            if last_rend < seed_end:
                return total + last_rend - last_start
            return total + seed_end - last_start

        if nonfactor_types:
            cov, prev_start = 0, _U32
            for j, ktype in enumerate(nonfactor_types):
                head(j, ktype)
                cov = covered(cov, prev_start)
                prev_start = last_start & _U32
            if cov > 0.5 * seed_len:              # :467
                self.met["rejected: covered by other motif sizes"] += 1
                return advanced

        if factor_types:                          # :471-526, per motif size (the maps default-insert 0)
            prev_starts = {f: -1 for f in factor_sizes}
            coverage = {f: 0 for f in factor_sizes}
            for j, ktype in enumerate(factor_types):
                head(j, ktype)
                prev_start = prev_starts.setdefault(last_mlen, 0) & _U32
                coverage[last_mlen] = covered(coverage.setdefault(last_mlen, 0), prev_start)
                prev_starts[last_mlen] = last_start
            for f in sorted(coverage):
                if coverage[f] >= 0.8 * seed_len:
                    m, seed_type = f, RANK_C
                    self.met["takes a factor's motif size"] += 1
                    for j, ktype in enumerate(factor_types):      # :511-522: entry j of the list, start and end as left above
                        lst = P if ktype == RANK_P else S if ktype == RANK_S else None
                        if lst is None:
                            continue
                        if j >= len(lst):         # D2
                            self.guards += 1
                            continue
                        if lst[j][2] == f:
                            if lst[j][:2] != [last_start, last_end]:
                                self.met["list-head write that moves an entry"] += 1
                            self.met["list-head write"] += 1
                            lst[j] = [last_start, last_end, lst[j][2], RANK_N]
                    break

        if seed_end > self.L - m:                 # :529
            seed_end = self.L - m
        A.append([seed_start, seed_end, m, seed_type])
        return advanced

    def run(self, calls):
        """calls: (pos, motif, start, end) in the window scan's order; pos == L marks the end-of-sequence calls,
        whose returned cursors the reference drops (parse_anchored_shiftxor.cpp:683-707)"""
        frm = (0, 0)
        for pos, m, start, end in calls:
            got = self.add(int(start), int(end), int(m), frm, RANK_A)
            if pos < self.L:
                frm = got
        return self
