"""How the path scales on a long homopolymer (every shift matches everywhere; it becomes ONE seed as long as the run,
whose alignment is quadratic in its length -- in the reference as here): product only, bounded sizes.
Usage: python tools/homopolymer_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ribbit_amd

rs = np.random.RandomState(1)
flank = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rs.randint(0, 4, size=500)])
for n in (5_000, 10_000, 20_000, 40_000):
    seq = flank + b"A" * n + flank
    with ribbit_amd.Scanner(2, 100) as sc:
        sc.load_record(seq)
        t = time.time()
        p, s, a = sc.processShiftXORsAnchored()
        dt = time.time() - t
        t = time.time()
        bed = sc.refine_bed("h")
        rt = time.time() - t
    print(f"homopolymer {n:6d}: seed lists {dt:6.2f} s ({len(p)} / {len(s)} / {len(a)} seeds), refinement + BED {rt:7.2f} s ({bed.count(chr(10))} rows)", flush=True)
