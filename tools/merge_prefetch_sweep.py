#!/usr/bin/env python3
"""The anchored merge's plane prefetch (parallel_merge.cpp, RIBBIT_MERGE_PREFETCH = how many calls ahead; 0 = none) on one
chromosome-sized record: the substitution and anchored stages are run once per setting on the same handle, and the stage's
RIBBIT_PROFILE line (parallel passes, the ranges' own times) is what to read.  The seed lists must not depend on the setting.
Usage (GPU box): python tools/merge_prefetch_sweep.py [bases] [distance ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RIBBIT_PROFILE"] = "1"
import numpy as np

import ribbit_amd
from ribbit_amd.simulate import simulate_sequence

bases = int(sys.argv[1]) if len(sys.argv) > 1 else 248_956_422
distances = [int(x) for x in sys.argv[2:]] or [0, 8, 0, 4, 16, 32, 8]
seq, _ = simulate_sequence(bases, 4, 2, 100)
first = None
with ribbit_amd.Scanner(2, 100) as sc:
    for d in distances:
        os.environ["RIBBIT_MERGE_PREFETCH"] = str(d)
        sc.load_record(seq)
        sc.processShiftXORsPerfect()
        t = time.perf_counter()
        p, s, a = sc.processShiftXORsAnchored()
        dt = time.perf_counter() - t
        print(f"prefetch {d} calls ahead: substitution + anchored stages {dt:.3f} s, anchored merge {sc.timing_ms(4):.0f} ms, {len(a)} anchored seeds", file=sys.stderr, flush=True)
        if first is None:
            first = a.copy()
        else:
            assert np.array_equal(first.view("<i4"), a.view("<i4"))
