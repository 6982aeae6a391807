#!/usr/bin/env python3
"""The anchored merge's handling of Q8's list-head writes (parallel_merge.cpp: ranges done again with the writes made, one more
parallel pass per head change a later range read) against the same stage made strictly in call order (RIBBIT_MERGE_FORCE_REDO=1)
on chromosome-sized records, where such writes do occur.  Usage (GPU box): python tools/merge_passes_check.py [bases] [seed ...]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import ribbit_amd
from ribbit_amd.simulate import simulate_sequence

bases = int(sys.argv[1]) if len(sys.argv) > 1 else 242_193_529
seeds = [int(x) for x in sys.argv[2:]] or [1004, 2004]
L = ribbit_amd.load_library()
for seed in seeds:
    seq, _ = simulate_sequence(bases, seed, 2, 100)
    got = {}
    for mode in ("parallel", "in order"):
        if mode == "in order":
            os.environ["RIBBIT_MERGE_FORCE_REDO"] = "1"
        else:
            os.environ.pop("RIBBIT_MERGE_FORCE_REDO", None)
        with ribbit_amd.Scanner(2, 100) as sc:
            sc.load_record(seq)
            sc.processShiftXORsPerfect()
            t = time.perf_counter()
            p, s, a = sc.processShiftXORsAnchored()
            d = sc.dispatch_seeds()
            dt = time.perf_counter() - t
            out = (C.c_int32 * 5)()
            L.ribbit_debug_last_merge(1, C.byref(out))
            got[mode] = (p.copy(), s.copy(), a.copy(), d.copy())
            print(f"seed {seed}, {bases} bases, {mode}: substitution + anchored stages {dt:.2f} s, anchored merge {sc.timing_ms(4):.0f} ms; ranges {out[0]}, "
                  f"done again {out[1]}, in order {out[2] & 1}, range runs {out[2] >> 1}, changing head writes {out[3]}, passes {out[4] >> 8}", flush=True)
    os.environ.pop("RIBBIT_MERGE_FORCE_REDO", None)
    same = all(np.array_equal(x.view("<i4"), y.view("<i4")) for x, y in zip(got["parallel"], got["in order"]))
    print(f"seed {seed}: lists identical to the in-order merge: {same}", flush=True)
    assert same
