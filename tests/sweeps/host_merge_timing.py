"""Time the product's host-side merges on this machine's CPU (no GPU): the oracle produces the call logs and
the composed planes of a simulated record, ribbit_host_replay_calls replays them.  Usage:
    python tests/sweeps/host_merge_timing.py [bases=2000000]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ribbit_amd
from ribbit_amd.simulate import simulate_sequence
from oracle_lib import LIST_ANCHORED, LIST_PERFECT, LIST_SUBST, Oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
seq, _ = simulate_sequence(n, seed=11)
t = time.time()
with Oracle(seq, 2, 100) as o:
    o.run_perfect(); pc = o.calls(LIST_PERFECT)
    o.run_subst(); sc = o.calls(LIST_SUBST)
    o.run_anchor_planes()
    xa, stride = ribbit_amd.pack_bit_planes([o.plane(m) for m in range(2, 101)], len(seq))
    o.run_anchored(); ac = o.calls(LIST_ANCHORED)
    want = o.seeds(LIST_ANCHORED)
print(f"oracle {time.time() - t:.1f}s  calls: perfect {len(pc)} subst {len(sc)} anchored {len(ac)}")
for label, args in (("perfect", (pc,)), ("perfect+subst", (pc, sc)), ("all three", (pc, sc, ac, xa, stride))):
    best = 1e9
    for _ in range(3):
        t = time.time()
        r = ribbit_amd.host_replay_calls(2, 100, seq, *args)
        best = min(best, time.time() - t)
    print(f"{label:14s} {best * 1e3:8.1f} ms")
assert np.array_equal(r["anchored"].view("<i4"), want.view("<i4"))
