#!/usr/bin/env python3
"""Per-stage timing of the scan on one GPU: kernel ms (HIP events), device state machine + read-back ms, host merge ms.
Usage: python tools/stage_timing.py [bases] [m_hi] [repeats]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ribbit_amd
from ribbit_amd.simulate import simulate_sequence

bases = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
m_hi = int(sys.argv[2]) if len(sys.argv) > 2 else 100
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
seq, _ = simulate_sequence(bases, 2, 2, m_hi)
with ribbit_amd.Scanner(2, m_hi) as sc:
    for rep in range(reps):
        print(f"--- pass {rep + 1} ({'cold: allocations included' if rep == 0 else 'warm'})")
        t = time.perf_counter(); sc.load_record(seq); print(f"load_record {1e3*(time.perf_counter()-t):.1f} ms (pack kernel {sc.timing_ms(0):.3f} ms)")
        t = time.perf_counter(); runs = sc.scan_perfect_runs(); dt = time.perf_counter() - t
        print(f"perfect: wall {1e3*dt:.1f} ms  kernel {sc.timing_ms(1):.3f} ms  gpu {sc.timing_ms(2):.3f} ms  events {sc.last_event_count()}  runs {len(runs)}")
        t = time.perf_counter(); seeds = sc.processShiftXORsPerfect(); print(f"perfect seeds {len(seeds)}  wall {1e3*(time.perf_counter()-t):.1f} ms")
        if m_hi > 990:
            continue
        t = time.perf_counter(); p, s, a = sc.processShiftXORsAnchored(); dt = time.perf_counter() - t
        d = sc.dispatch_seeds()
        print(f"substitution + anchored stages: wall {1e3*dt:.1f} ms  anchored kernel {sc.timing_ms(1):.3f} ms  host merges {sc.timing_ms(5):.1f} + {sc.timing_ms(4):.1f} ms  "
              f"seeds {len(s)} + {len(a)}  dispatch {len(d)}  guards {sc.guard_hits()}")
    # the full call lists (the parity entry points), warm
    sc.load_record(seq)
    t = time.perf_counter(); calls = sc.subst_calls(); print(f"subst_calls (full list): wall {1e3*(time.perf_counter()-t):.1f} ms  calls {len(calls)}")
    t = time.perf_counter(); calls = sc.anchored_calls(); print(f"anchored_calls (full list): wall {1e3*(time.perf_counter()-t):.1f} ms  calls {len(calls)}")
