#!/usr/bin/env python3
"""Per-stage timing of the scan on one GPU: kernel ms (HIP events), events, host post-processing ms.
Usage: python tools/stage_timing.py [bases] [m_hi]"""
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
    t = time.perf_counter(); sc.load_record(seq); print(f"load_record {1e3*(time.perf_counter()-t):.1f} ms (pack kernel {sc.timing_ms(0):.3f} ms)")
    for rep in range(2):
        t = time.perf_counter(); runs = sc.scan_perfect_runs(); dt = time.perf_counter() - t
        print(f"perfect: wall {1e3*dt:.1f} ms  kernel {sc.timing_ms(1):.3f} ms  gpu {sc.timing_ms(2):.3f} ms  host {sc.timing_ms(3):.1f} ms  events {sc.last_event_count()}  runs {len(runs)}")
    t = time.perf_counter(); seeds = sc.processShiftXORsPerfect(); print(f"perfect seeds {len(seeds)}  wall {1e3*(time.perf_counter()-t):.1f} ms")
    t = time.perf_counter(); calls = sc.subst_calls(); dt = time.perf_counter() - t
    print(f"subst: wall {1e3*dt:.1f} ms  kernel {sc.timing_ms(1):.3f} ms  gpu {sc.timing_ms(2):.3f} ms  host(fsm+sort) {sc.timing_ms(3):.1f} ms  events {sc.last_event_count()}  calls {len(calls)}")
    t = time.perf_counter(); p, s = sc.processShiftXORswithSubstitutions(); print(f"subst seeds {len(s)}  merge wall {1e3*(time.perf_counter()-t):.1f} ms")
    if m_hi <= 110:
        t = time.perf_counter(); calls = sc.anchored_calls(); dt = time.perf_counter() - t
        print(f"anchored: wall {1e3*dt:.1f} ms  kernel {sc.timing_ms(1):.3f} ms  gpu {sc.timing_ms(2):.3f} ms  host(fsm+sort) {sc.timing_ms(3):.1f} ms  events {sc.last_event_count()}  calls {len(calls)}")
        t = time.perf_counter(); p, s, a = sc.processShiftXORsAnchored(); d = sc.dispatch_seeds()
        print(f"anchored seeds {len(a)} dispatch {len(d)}  merge wall {1e3*(time.perf_counter()-t):.1f} ms  guards {sc.guard_hits()}")
