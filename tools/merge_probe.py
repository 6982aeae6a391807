#!/usr/bin/env python3
"""Profile lines (RIBBIT_PROFILE) of the scans and merges of one simulated record, three passes: what the anchored stage's merge
does on the device and on the host threads.  usage: merge_probe.py BASES [MAX_MOTIF] [passes]"""
import os
import sys
import time

os.environ.setdefault("RIBBIT_PROFILE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ribbit_amd  # noqa: E402
from ribbit_amd import simulate  # noqa: E402

bases = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000_000
m_hi = int(sys.argv[2]) if len(sys.argv) > 2 else 100
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
seq = simulate.grch38_shaped_record(1, bases, 2, m_hi)
with ribbit_amd.Scanner(2, m_hi) as sc:
    for p in range(passes):
        t = time.time()
        sc.load_record(seq)
        perfect, subst, anchored = sc.processShiftXORsAnchored()
        print(f"pass {p}: {time.time() - t:.3f} s, {len(perfect)} + {len(subst)} + {len(anchored)} seeds, device merge {ribbit_amd.last_device_merge()}", file=sys.stderr, flush=True)
