#!/usr/bin/env python3
"""Size distribution of a record's first-level alignment jobs: how much of the (query x reference) work lies in each class.
Usage: python tools/job_sizes.py [bases]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import ribbit_amd
from ribbit_amd.simulate import simulate_sequence

bases = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
seq, _ = simulate_sequence(bases, 2, 2, 100)
with ribbit_amd.Scanner(2, 100) as sc:
    sc.load_record(seq)
    jobs, _pool = sc.refine_jobs()
q, r = jobs["query_length"].astype(np.int64), jobs["ppr_length"].astype(np.int64)
work = q * r
print(f"{len(jobs)} jobs, total work {work.sum() / 1e9:.1f} G cells")
for lo, hi in ((0, 128), (128, 512), (512, 2048), (2048, 4096), (4096, 8192), (8192, 1 << 30)):
    m = (q > lo) & (q <= hi)
    print(f"query {lo + 1:>5}..{hi if hi < 1 << 30 else 'inf':>5}: {int(m.sum()):>8} jobs, {100 * work[m].sum() / work.sum():5.1f} % of the cells, longest reference {int(r[m].max()) if m.any() else 0}")
