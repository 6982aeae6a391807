#!/usr/bin/env python3
"""Host timing of the range-parallel seed-list merges without a GPU: the oracle (test infrastructure, which is why this
script lives under tests/) makes the three call lists of one generator record once and caches them in /tmp; the product's
host merges (ribbit_host_replay_calls -> parallel_merge.cpp) are then run on them repeatedly with RIBBIT_PROFILE lines on.

    python tests/sweeps/merge_host_timing.py [--mbp 20] [--seed 4] [--threads 8] [--repeat 3]

What the GPU path hands the merges is the same kept-call list, so per-call costs measured here carry over; the page-fault
and memory-bandwidth share grows with the record (32.8 M anchored seeds = 524 MB at chromosome-1 size)."""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbp", type=float, default=20.0)
    ap.add_argument("--seed", type=int, default=4)
    ap.add_argument("--max-motif", type=int, default=100)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--recompute-planes", action="store_true", help="do not hand the composed planes over: the merges recompute the slices they read")
    a = ap.parse_args()
    os.environ["RIBBIT_THREADS"] = str(a.threads)
    os.environ["RIBBIT_PROFILE"] = "1"
    import ribbit_amd
    from oracle_lib import LIST_ANCHORED, LIST_PERFECT, LIST_SUBST, Oracle
    from ribbit_amd.simulate import simulate_sequence

    bases = int(a.mbp * 1e6)
    cache = f"/tmp/merge_timing_xa_{bases}_{a.seed}_{a.max_motif}.npz"
    seq, _ = simulate_sequence(bases, a.seed, 2, a.max_motif)
    if os.path.exists(cache):
        g = np.load(cache)
        calls = [g["p"], g["s"], g["a"]]
        want_anchored = int(g["n_anchored"])
        xa, stride = g["xa"], int(g["stride"])
    else:
        t = time.perf_counter()
        with Oracle(seq, 2, a.max_motif) as o:
            o.run_all()
            calls = [o.calls(w).copy() for w in (LIST_PERFECT, LIST_SUBST, LIST_ANCHORED)]
            want_anchored = len(o.seeds(LIST_ANCHORED))
            # the composed planes XA_m as stored planes (what the GPU path copies back for the merges), one motif at a time
            stride = (bases // 32 + 1 + 7) // 8 * 8 + 16
            xa = np.zeros((a.max_motif - 1, stride), dtype="<u4")
            for m in range(2, a.max_motif + 1):
                xa[m - 2] = ribbit_amd.pack_bit_planes([o.plane(m)], bases)[0][0]
        np.savez(cache, p=calls[0], s=calls[1], a=calls[2], n_anchored=want_anchored, xa=xa, stride=stride)
        print(f"oracle: {time.perf_counter() - t:.1f} s", file=sys.stderr)
    print(f"{bases} bases: {len(calls[0])} perfect, {len(calls[1])} substitution, {len(calls[2])} anchored calls", file=sys.stderr)
    for k in range(a.repeat):
        t = time.perf_counter()
        r = ribbit_amd.host_replay_calls(2, a.max_motif, seq, *calls, xa=None if a.recompute_planes else xa, xa_stride=stride)
        print(f"run {k}: host merges of all three stages {1e3 * (time.perf_counter() - t):.1f} ms, {len(r['anchored'])} anchored seeds "
              f"({'as the oracle' if len(r['anchored']) == want_anchored else 'NOT the oracle count'})", file=sys.stderr, flush=True)


if __name__ == "__main__":
    main()
