"""Whole FASTA->BED path on ONE record of 64 Mbp (BASELINE configs[2]'s shape: N blocks, lower case, motifs 2..100;
RIBBIT_TEST_BASES sets another size) against the oracle pipeline run on the same whole record, byte for byte.
The oracle needs ~5 s per Mbp on one core (5-6 minutes here), in a spawned process that never touches the GPU; the GPU
path takes seconds (for a record this large it includes the batched GPU alignment path, which is on by default from
two million dispatched seeds: the test checks that it was taken).  A line a minute goes to the terminal and to gpurun_out/ so that the wait is not taken for a hang."""
import multiprocessing
import os
import time

import pytest

import ribbit_amd
import segments

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_whole_path_on_a_64_megabase_record_matches_oracle(capsys):
    total = int(os.environ.get("RIBBIT_TEST_BASES", "64000000"))
    with multiprocessing.get_context("spawn").Pool(1) as pool:
        pending = pool.apply_async(segments.oracle_bed_of_simulated_record, ((total, 500, "chr"),))
        record = segments.simulated_record(total, 500)
        assert len(record) >= total
        t0 = time.time()
        before = ribbit_amd.alignment_counters()
        with ribbit_amd.Scanner(2, 100) as sc:
            sc.load_record(record)
            got = sc.refine_bed("chr")
        gpu_s = time.time() - t0
        made, passes, paths = (b - a for a, b in zip(before, ribbit_amd.alignment_counters()))
        if total >= 40_000_000 and "RIBBIT_GPU_SSW" not in os.environ:
            # a record of this size (millions of dispatched seeds) has its alignments batched on the GPU by default
            assert passes > 0.9 * made and paths > 0.9 * made, (made, passes, paths)
        del record
        progress = os.path.join(ROOT, "gpurun_out", "large_test_progress.log")
        while not pending.ready():
            pending.wait(60)
            line = f"[whole-path test, {total / 1e6:.0f} Mbp] GPU path done in {gpu_s:.1f} s ({got.count(chr(10))} rows); oracle running for {time.time() - t0:.0f} s"
            with capsys.disabled():
                print(line, flush=True)
            if os.path.isdir(os.path.dirname(progress)):
                with open(progress, "a") as f:
                    f.write(line + "\n")
        want = pending.get()
    assert want.count("\n") > 9000 * (total // 1_000_000)
    if got != want:                      # name the first differing row instead of dumping two 60-MB strings
        g, w = got.splitlines(), want.splitlines()
        k = next((i for i, (a, b) in enumerate(zip(g, w)) if a != b), min(len(g), len(w)))
        pytest.fail(f"rows {len(g)} vs {len(w)}; first difference at row {k}: {g[k] if k < len(g) else None!r} vs {w[k] if k < len(w) else None!r}")


def test_list_head_writes_on_a_chromosome_sized_record_keep_the_merge_parallel_and_exact():
    """Q8's writes to list heads (parse_anchored_shiftxor.cpp:511-522) that change an entry are rare per megabase and routine
    per chromosome: generator seed 1004 at chromosome 2's size makes one.  The merge must stay parallel (a second pass over the
    ranges behind the change, not the whole stage on one thread) and give exactly the lists of the same stage made strictly in
    call order (RIBBIT_MERGE_FORCE_REDO=1: the reference's order of calls, one after the other)."""
    import ctypes as C

    import numpy as np
    from ribbit_amd.simulate import simulate_sequence
    bases = int(os.environ.get("RIBBIT_TEST_HEAD_BASES", "242193529"))
    seq, _ = simulate_sequence(bases, 1004, 2, 100)
    lists, stats = {}, {}
    for mode in ("parallel", "in order"):
        if mode == "in order":
            os.environ["RIBBIT_MERGE_FORCE_REDO"] = "1"
        try:
            with ribbit_amd.Scanner(2, 100) as sc:
                sc.load_record(seq)
                p, s, a = sc.processShiftXORsAnchored()
                lists[mode] = (p.copy(), s.copy(), a.copy(), sc.dispatch_seeds().copy())
                out = (C.c_int32 * 5)()
                ribbit_amd.load_library().ribbit_debug_last_merge(1, C.byref(out))
                stats[mode] = [int(x) for x in out]
        finally:
            os.environ.pop("RIBBIT_MERGE_FORCE_REDO", None)
    assert stats["in order"][2] & 1 == 1 and stats["parallel"][2] & 1 == 0, stats
    if bases == 242193529:
        assert stats["parallel"][3] >= 1 and (stats["parallel"][4] >> 8) >= 2, stats          # the write is there, and cost one more pass
    for x, y in zip(lists["parallel"], lists["in order"]):
        assert np.array_equal(x.view("<i4"), y.view("<i4"))
