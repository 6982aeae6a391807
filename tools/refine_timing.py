#!/usr/bin/env python3
"""Refinement (dispatch list -> BED text) timing on one GPU.  Usage: python tools/refine_timing.py [bases] [m_hi]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ribbit_amd
from ribbit_amd.simulate import simulate_sequence

bases = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
m_hi = int(sys.argv[2]) if len(sys.argv) > 2 else 100
seq, _ = simulate_sequence(bases, 2, 2, m_hi)
with ribbit_amd.Scanner(2, m_hi) as sc:
    for rep in range(2):
        sc.load_record(seq)
        t = time.perf_counter(); sc.processShiftXORsAnchored(); d = sc.dispatch_seeds(); t1 = time.perf_counter()
        # the text as the C ABI returns it (a view of the library's buffer), as bench.py's chr1 leg takes it: the mirror's copy
        # into a Python str is 0.2 s at chromosome size and no part of the path
        bed = sc.refine_bed_view("x"); t2 = time.perf_counter()
        rows = int((bed == 10).sum())
        print(f"pass {rep + 1}: scans + merges {1e3 * (t1 - t):.1f} ms, refinement + BED {1e3 * (t2 - t1):.1f} ms, {len(d)} seeds dispatched, {rows} rows", flush=True)
