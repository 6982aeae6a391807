#!/usr/bin/env python3
"""Kernel-only timing (HIP events) of scan_perfect_kernel on the bench workload and on random bases.
Usage: python tools/perfect_probe.py [bases] [reps]   (RIBBIT_HIP_LIBRARY selects a variant build)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import ribbit_amd
from ribbit_amd.simulate import random_sequence, simulate_sequence

bases = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
for label, seq in (("simulated", simulate_sequence(bases, 2, 2, 100)[0]), ("random", random_sequence(bases, 3))):
    with ribbit_amd.Scanner(2, 100) as sc:
        sc.load_record(seq)
        k = []
        for rep in range(reps + 2):
            try:
                sc.scan_perfect_runs()
            except ribbit_amd.RibbitHipError:       # ablated builds drop events
                pass
            k.append(sc.timing_ms(1))
        k = k[2:]
        print(f"{os.environ.get('RIBBIT_HIP_LIBRARY', 'product')}: {label}: scan_perfect_kernel {np.median(k):.4f} ms  "
              f"({bases / np.median(k) / 1e6:.1f} Gbases/s)  events {sc.last_event_count()}", flush=True)
