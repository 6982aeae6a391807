"""Regenerates tests/golden/*.npz from the CPU oracle.

These fixtures are NOT reference outputs: the reference ships no golden vectors and cannot be
built or run in this image (DESIGN.md "Oracle"), so parity is UNPINNED.
The files lock the oracle's current behaviour (regression) and give the GPU parity tests inputs
with known-good expected outputs that travel to the GPU box without /root/reference.

Usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from cases import edge_cases, simulated_cases  # noqa: E402
from oracle_lib import LIST_ANCHORED, LIST_PERFECT, LIST_SUBST, Oracle  # noqa: E402


def snapshot(seq, m_lo, m_hi):
    with Oracle(seq, m_lo, m_hi) as o:
        o.run_perfect()
        perfect_after_p = o.seeds(LIST_PERFECT)
        o.run_subst()
        perfect_after_s = o.seeds(LIST_PERFECT)
        subst_after_s = o.seeds(LIST_SUBST)
        o.run_anchor_planes()
        o.run_anchored()
        o.run_dispatch()
        return dict(
            seq=np.frombuffer(seq, dtype=np.uint8), m_lo=m_lo, m_hi=m_hi,
            perfect_calls=o.calls(LIST_PERFECT), subst_calls=o.calls(LIST_SUBST), anchored_calls=o.calls(LIST_ANCHORED),
            perfect_after_p=perfect_after_p, perfect_after_s=perfect_after_s, subst_after_s=subst_after_s,
            perfect=o.seeds(LIST_PERFECT), subst=o.seeds(LIST_SUBST), anchored=o.seeds(LIST_ANCHORED),
            dispatch=o.dispatch(), guard_hits=o.guard_hits())


def main():
    for name, seq, m_lo, m_hi in edge_cases() + simulated_cases():
        snap = snapshot(seq, m_lo, m_hi)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **snap)
        print(f"{name}: L={len(seq)} m={m_lo}..{m_hi} perfect={len(snap['perfect'])} subst={len(snap['subst'])} "
              f"anchored={len(snap['anchored'])} dispatch={len(snap['dispatch'])} guards={snap['guard_hits']}")


if __name__ == "__main__":
    main()
