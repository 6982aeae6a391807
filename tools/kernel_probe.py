#!/usr/bin/env python3
"""Kernel-only timings (HIP events) of the three scan kernels on simulated vs uniform-random input."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ribbit_amd
from ribbit_amd.simulate import simulate_sequence, random_sequence
bases = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
for label, seq in (("simulated", simulate_sequence(bases, 2, 2, 100)[0]), ("random", random_sequence(bases, 3))):
    with ribbit_amd.Scanner(2, 100) as sc:
        sc.load_record(seq)
        for rep in range(3):
            sc.scan_perfect_runs()
        k = [];
        for rep in range(5):
            sc.scan_perfect_runs(); k.append(sc.timing_ms(1))
        print(f"{label}: perfect kernel {np.median(k):.3f} ms  events {sc.last_event_count()}  -> {bases/np.median(k)/1e6:.1f} Gbases/s")
        if bases <= 20_000_000:
            sc.subst_calls(); print(f"{label}: window<1> kernel {sc.timing_ms(1):.3f} ms events {sc.last_event_count()}")
            sc.anchored_calls(); print(f"{label}: anchored kernel {sc.timing_ms(1):.3f} ms events {sc.last_event_count()}")
