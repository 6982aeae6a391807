"""BASELINE.json configs[4] as stated: `-m 2 -M 500` (MAXIMUM_SHIFT = 502, ribbit.cpp:240-243; 499 composed planes; motifs
beyond 128 bases wrap in the reference's uint256_t, parse_seed.cpp:246-253, Q11) on ONE record of RIBBIT_TEST_M500_BASES
bases (default 16 Mbp: generator motifs 2..500, blocks of N, lower case) against the oracle pipeline on the same record:
the three seed lists, the dispatch order and the BED text, byte for byte.  The oracle needs ~12 s per Mbp on one core of
the GPU box at this motif range (five times the planes and calls of -M 100) and runs in a spawned process that never
touches the GPU; a line a minute goes to the terminal and to gpurun_out/ so that the wait is not taken for a hang."""
import hashlib
import multiprocessing
import os
import sys
import time

import numpy as np
import pytest

import ribbit_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M_LO, M_HI = 2, 500


def m500_record(bases: int) -> bytes:
    from ribbit_amd.simulate import m500_record as record
    return record(bases)


def _digest(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).tobytes()).hexdigest()


def oracle_side(bases: int):
    """worker (no GPU, own process)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import LIST_ANCHORED, LIST_PERFECT, LIST_SUBST, Oracle
    with Oracle(m500_record(bases), M_LO, M_HI) as o:
        o.run_all()
        lists = {k: (_digest(o.seeds(w)), int(len(o.seeds(w)))) for k, w in (("perfect", LIST_PERFECT), ("subst", LIST_SUBST), ("anchored", LIST_ANCHORED))}
        lists["dispatch"] = (_digest(o.dispatch()), int(len(o.dispatch())))
        return lists, o.guard_hits(), o.refine_bed("read500")


def test_whole_path_at_M_500_on_a_sixteen_megabase_record_matches_oracle(capsys):
    total = int(os.environ.get("RIBBIT_TEST_M500_BASES", "16000000"))
    with multiprocessing.get_context("spawn").Pool(1) as pool:
        pending = pool.apply_async(oracle_side, (total,))
        record = m500_record(total)
        t0 = time.time()
        with ribbit_amd.Scanner(M_LO, M_HI) as sc:
            sc.load_record(record)
            p, s, a = sc.processShiftXORsAnchored()
            got_lists = {"perfect": (_digest(p), len(p)), "subst": (_digest(s), len(s)), "anchored": (_digest(a), len(a))}
            d = sc.dispatch_seeds()
            got_lists["dispatch"] = (_digest(d), len(d))
            longest = int((a["mlen"]).max()) if len(a) else 0
            got = sc.refine_bed("read500")
        gpu_s = time.time() - t0
        del record
        progress = os.path.join(ROOT, "gpurun_out", "m500_test_progress.log")
        while not pending.ready():
            pending.wait(60)
            line = f"[-M 500 test, {total / 1e6:.0f} Mbp] GPU path done in {gpu_s:.1f} s ({got.count(chr(10))} rows); oracle running for {time.time() - t0:.0f} s"
            with capsys.disabled():
                print(line, flush=True)
            if os.path.isdir(os.path.dirname(progress)):
                with open(progress, "a") as f:
                    f.write(line + "\n")
        want_lists, _, want = pending.get()
    assert longest > 128, longest          # the record does exercise motifs beyond the uint256_t's 128 bases (Q11)
    for k in ("perfect", "subst", "anchored", "dispatch"):
        assert (got_lists[k][0], int(got_lists[k][1])) == want_lists[k], (k, got_lists[k], want_lists[k])
    assert want.count("\n") > 3000 * (total // 1_000_000)
    if got != want:
        g, w = got.splitlines(), want.splitlines()
        k = next((i for i, (x, y) in enumerate(zip(g, w)) if x != y), min(len(g), len(w)))
        pytest.fail(f"rows {len(g)} vs {len(w)}; first difference at row {k}: {g[k] if k < len(g) else None!r} vs {w[k] if k < len(w) else None!r}")
