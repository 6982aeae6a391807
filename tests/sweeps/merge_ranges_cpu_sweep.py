"""One-off wide fuzz of the range-parallel seed-list merges on the CPU (no GPU needed): the oracle's call lists of fuzz records
replayed through ribbit_host_replay_calls with ranges of 1 and 3 calls -- the host threads' ranges (`plain`) or the ranges of the
GPU's pass of the anchored merge with a device that cannot run (`dev`: RIBBIT_MERGE_DEVICE_RANGES) -- against the oracle's lists,
dispatch order and guard count.  Round 4 found fuzz seed 430991 this way (a list-head write that only mattered inside its own
range); 196,000 records since without a mismatch.
Usage: python tests/sweeps/merge_ranges_cpu_sweep.py <lo> <hi> plain|dev [scale]     (run several ranges of seeds side by side)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ribbit_amd
from fuzz import fuzz_case
from oracle_lib import LIST_ANCHORED, LIST_PERFECT, LIST_SUBST, Oracle

lo, hi, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
scale = int(sys.argv[4]) if len(sys.argv) > 4 else 1
lib = ribbit_amd.load_library()
os.environ["RIBBIT_THREADS"] = "2"
bad = 0
for seed in range(lo, hi):
    seq, m_lo, m_hi = fuzz_case(seed, scale)
    with Oracle(seq, m_lo, m_hi) as o:
        o.run_all()
        calls = [o.calls(k) for k in (LIST_PERFECT, LIST_SUBST, LIST_ANCHORED)]
        want, order, guards = o.seeds(LIST_ANCHORED).view("<i4"), o.dispatch().view("<i4"), o.guard_hits()
        for calls_per_range in (1, 3):
            lib.ribbit_debug_set_merge_min_range(calls_per_range)
            if mode == "dev":
                os.environ["RIBBIT_MERGE_DEVICE_RANGES"] = str(calls_per_range)
            r = ribbit_amd.host_replay_calls(m_lo, m_hi, seq, *calls)
            if not (np.array_equal(r["anchored"].view("<i4"), want) and np.array_equal(r["dispatch"].view("<i4"), order) and r["guard_hits"] == guards):
                bad += 1
                print(f"MISMATCH seed {seed}: {len(seq)} bases -m {m_lo} -M {m_hi}, ranges of {calls_per_range} ({mode})", flush=True)
print(f"seeds [{lo}, {hi}) {mode}: {bad} mismatches")
