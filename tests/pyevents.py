"""Test-side numpy restatement of what the scan kernels emit (device events), built from the oracle's planes:
used to build the perfect-stage run records of the chunk-sharded CPU tests (tests/test_sharded.py) and in the streak tests; incl. the chunk-sharded
exchange under gloo.  Test-only."""
import numpy as np


def _pack(pos, m, kind):
    return pos.astype(np.uint64) | (np.uint64(m) << np.uint64(32)) | (kind.astype(np.uint64) << np.uint64(48))


def perfect_events(oracle, m_lo, m_hi):
    """per motif: START / END of every run of X_m & ~N with length >= min(c1, 32) (kernels.hip scan_perfect_kernel)"""
    nmask = oracle.nmask().astype(np.int8)
    L = len(nmask)
    out, counts = [], []
    for m in range(m_lo, m_hi + 1):
        sp = min(12 - m if m <= 6 else m, 32)
        y = (oracle.plane(m).astype(np.int8) & (1 - nmask)) if L else np.zeros(0, np.int8)
        d = np.diff(np.concatenate(([0], y, [0])))
        s, e = np.flatnonzero(d == 1), np.flatnonzero(d == -1)
        keep = (e - s) >= sp
        s, e = s[keep], e[keep]
        kind_e = np.where(e >= L, 3, np.where(nmask[np.minimum(e, L - 1)] == 1, 2, 1)) if len(e) else np.zeros(0, np.int64)
        pos = np.concatenate((s, e)); kind = np.concatenate((np.zeros(len(s), np.int64), kind_e))
        order = np.argsort(pos, kind="stable")
        out.append(_pack(pos[order], m, kind[order])); counts.append(len(pos))
    return (np.concatenate(out) if out else np.zeros(0, np.uint64)), np.array(counts, dtype=np.uint64)


def window_events(oracle, m_lo, m_hi, allowed):
    """per motif: START / END of every pass-streak of the 8-window scan on the oracle's CURRENT planes"""
    nmask = oracle.nmask().astype(np.int32)
    L = len(nmask)
    out, counts = [], []
    nq = max(L - 7, 0)
    csn = np.concatenate(([0], np.cumsum(nmask)))
    evalw = (csn[8:8 + nq] - csn[:nq]) == 0 if nq else np.zeros(0, bool)
    for m in range(m_lo, m_hi + 1):
        mism = 1 - oracle.plane(m).astype(np.int32)
        cs = np.concatenate(([0], np.cumsum(mism)))
        bad = (cs[8:8 + nq] - cs[:nq]) > allowed if nq else np.zeros(0, bool)
        p = np.zeros(L + 2, np.int8)                        # pass[q] for q = 0..L (0 beyond L-8)
        p[:nq] = evalw & ~bad
        prev = np.concatenate(([0], p[:-1]))
        starts = np.flatnonzero((p == 1) & (prev == 0))
        ends = np.flatnonzero((p == 0) & (prev == 1))
        ev_at = np.zeros(L + 2, bool); ev_at[:nq] = evalw
        kind_e = np.where(ev_at[ends], 1, np.where(ends + 7 >= L, 3, 2)) if len(ends) else np.zeros(0, np.int64)
        pos = np.concatenate((starts, ends)); kind = np.concatenate((np.zeros(len(starts), np.int64), kind_e))
        order = np.argsort(pos, kind="stable")
        out.append(_pack(pos[order], m, kind[order])); counts.append(len(pos))
    return (np.concatenate(out) if out else np.zeros(0, np.uint64)), np.array(counts, dtype=np.uint64)


def split_events(ev, counts, own_lo, own_hi):
    """the events of one part: positions in [own_lo, own_hi), motif-major"""
    out, cnt = [], []
    off = 0
    for c in counts:
        seg = ev[off:off + int(c)]; off += int(c)
        pos = (seg & np.uint64(0xFFFFFFFF)).astype(np.int64)
        sel = seg[(pos >= own_lo) & (pos < own_hi)]
        out.append(sel); cnt.append(len(sel))
    return (np.concatenate(out) if out else np.zeros(0, np.uint64)), np.array(cnt, dtype=np.uint64)
