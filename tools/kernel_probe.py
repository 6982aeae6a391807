#!/usr/bin/env python3
"""Kernel-only timings (HIP events) of the three scan kernels on one record.  Usage: python tools/kernel_probe.py [bases] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import ribbit_amd
from ribbit_amd.simulate import random_sequence, simulate_sequence

bases = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for label, seq in (("simulated", simulate_sequence(bases, 2, 2, 100)[0]), ("random", random_sequence(bases, 3))):
    with ribbit_amd.Scanner(2, 100) as sc:
        sc.load_record(seq)
        for rep in range(2):
            sc.scan_perfect_runs()
        k = []
        for rep in range(reps):
            sc.scan_perfect_runs(); k.append(sc.timing_ms(1))
        print(f"{label}: scan_perfect_kernel {np.median(k):.4f} ms  ({bases / np.median(k) / 1e6:.1f} Gbases/s)  events {sc.last_event_count()}")
        w, a = [], []
        for rep in range(max(2, reps // 2)):
            sc.load_record(seq)
            try:
                sc.processShiftXORsAnchored()
            except ribbit_amd.RibbitHipError as e:      # (experimental kernel builds that drop events)
                print("  stage failed:", str(e)[:80])
                try:
                    sc.anchored_calls()
                except ribbit_amd.RibbitHipError:
                    pass
            w.append(sc.timing_ms(6)); a.append(sc.timing_ms(7))
        print(f"{label}: scan_window_kernel<1> {np.median(w):.4f} ms ({bases / np.median(w) / 1e6:.1f} Gbases/s)   "
              f"scan_anchored_kernel {np.median(a):.4f} ms ({bases / np.median(a) / 1e6:.1f} Gbases/s)", flush=True)
