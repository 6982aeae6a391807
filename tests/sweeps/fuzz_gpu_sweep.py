"""One-off wide fuzz on the GPU box: whole path against the oracle for seeds [lo, hi) of tests/fuzz.py.
Usage: python tests/sweeps/fuzz_gpu_sweep.py <lo> <hi> [scale [m_lo m_hi]]  (prints mismatching seeds; progress every 200 cases;
scale stretches every record: 16 puts most of them over several 16-kb kernel tiles; m_lo m_hi: every record at THIS motif range
instead of the seed's own -- `1 2 500` are the "M500 rows" of round 4: the fuzz records at BASELINE.json configs[4]'s -m 2 -M 500,
with a few long-motif pieces (units of 150-480 bases) mixed into each so that the range's upper end has something to find)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ribbit_amd
from fuzz import fuzz_case
from oracle_lib import LIST_ANCHORED, LIST_PERFECT, LIST_SUBST, Oracle

lo, hi = int(sys.argv[1]), int(sys.argv[2])
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1
fixed_range = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else None


def long_motif_pieces(seed):
    rs = np.random.RandomState(1_000_003 + seed)
    out = b""
    for _ in range(int(rs.randint(1, 4))):
        unit = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rs.randint(0, 4, size=int(rs.randint(150, 481)))])
        body = bytearray(unit * int(rs.randint(2, 6)))
        for k in rs.choice(len(body), len(body) // int(rs.choice([15, 40, 200])), replace=False):
            body[k] = b"ACGT"[rs.randint(4)]
        out += bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rs.randint(0, 4, size=int(rs.randint(20, 300)))]) + bytes(body)
    return out


bad, t0 = 0, time.time()
for seed in range(lo, hi):
    seq, m_lo, m_hi = fuzz_case(seed, scale)
    if fixed_range:
        m_lo, m_hi = fixed_range
        if m_hi > 140:
            seq = seq[:len(seq) // 2] + long_motif_pieces(seed) + seq[len(seq) // 2:]
    with ribbit_amd.Scanner(m_lo, m_hi) as sc, Oracle(seq, m_lo, m_hi) as o:
        sc.load_record(seq)
        o.run_all()
        if seed % 2:                     # odd seeds: seed lists first (compact anchored calls), the call lists afterwards
            sc.processShiftXORsAnchored()
        ok = np.array_equal(sc.perfect_calls().view("<i4"), o.calls(LIST_PERFECT).view("<i4"))
        ok &= np.array_equal(sc.subst_calls().view("<i4"), o.calls(LIST_SUBST).view("<i4"))
        ok &= np.array_equal(sc.anchored_calls().view("<i4"), o.calls(LIST_ANCHORED).view("<i4"))
        p, s, a = sc.processShiftXORsAnchored()
        ok &= np.array_equal(p.view("<i4"), o.seeds(LIST_PERFECT).view("<i4")) and np.array_equal(s.view("<i4"), o.seeds(LIST_SUBST).view("<i4"))
        ok &= np.array_equal(a.view("<i4"), o.seeds(LIST_ANCHORED).view("<i4")) and np.array_equal(sc.dispatch_seeds().view("<i4"), o.dispatch().view("<i4"))
        ok &= sc.guard_hits() == o.guard_hits() and sc.refine_bed("fz") == o.refine_bed("fz")
    if not ok:
        bad += 1
        print(f"MISMATCH seed {seed}: {len(seq)} bases -m {m_lo} -M {m_hi}", flush=True)
    if (seed - lo) % (200 if scale == 1 else 20) == (199 if scale == 1 else 19):
        print(f"... {seed - lo + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"seeds [{lo}, {hi}): {bad} mismatches")
