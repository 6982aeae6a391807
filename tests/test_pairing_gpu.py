"""Fault injection into the device-side pairing of the perfect scan's events (pair_* kernels): a
well-formed stream pairs into ordered runs; malformed ones raise the corresponding flag instead of producing runs
silently.  Events: pos | mlen << 32 | kind << 48, kind 0 START, 1/2/3 END (mismatch / N / end of sequence)."""
import numpy as np
import pytest

import ribbit_amd

pytestmark = pytest.mark.gpu
S, E0, EN, EE = 0, 1, 2, 3
TILE = 16384


def ev(pos, m, kind):
    return np.uint64(pos) | (np.uint64(m) << np.uint64(32)) | (np.uint64(kind) << np.uint64(48))


def run(events, length=5 * TILE):
    with ribbit_amd.Scanner(2, 20) as sc:
        return sc.debug_pair_events(np.array(events, dtype="<u8"), length)


def test_well_formed_stream_pairs_into_runs_ordered_by_motif_and_start():
    # chunks in arbitrary (arrival) order; one run crosses two empty tiles; one ends at the end of the sequence
    events = [ev(40000, 7, S), ev(40100, 7, E0),                                     # motif 7, tile 2
              ev(10, 3, S), ev(60, 3, EN), ev(200, 3, S), ev(900, 3, E0),            # motif 3, tile 0
              ev(100, 7, S), ev(150, 7, E0), ev(16000, 7, S),                        # motif 7, tile 0 (run open at its end)
              ev(5 * TILE, 3, EE),                                                   # motif 3, tile 5: closes the run from tile 1
              ev(70000, 7, E0),                                                      # motif 7, tile 4: closes 16000 (tiles 1..3 empty of starts)
              ev(20000, 3, S)]                                                       # motif 3, tile 1
    # motif 7: tile 2 has its own complete run in between -> the run from 16000 would have to skip it: make it legal
    events = [e for e in events if int(e) & 0xFFFFFFFF not in (40000, 40100)]
    runs, flags = run(events)
    assert flags == 0
    got = [tuple(int(x) for x in r) for r in runs]
    assert got == [(10, 60, 3, 1), (200, 900, 3, 0), (20000, 5 * TILE, 3, 2), (100, 150, 7, 0), (16000, 70000, 7, 0)]


@pytest.mark.parametrize("events,bit,what", [
    ([ev(10, 3, S), ev(60, 3, E0), ev(90, 3, E0)], 4, "two ENDs in a row inside a chunk"),
    ([ev(10, 3, S), ev(40, 3, S), ev(60, 3, E0)], 4, "two STARTs in a row inside a chunk"),
    ([ev(60, 3, E0), ev(100, 3, S), ev(200, 3, E0)], 4, "END with no run open (first event of the motif)"),
    ([ev(10, 3, S), ev(60, 3, E0), ev(20000, 3, E0)], 4, "chunk opens with an END but the previous chunk closed its run"),
    ([ev(10, 3, S), ev(60, 3, E0), ev(100, 3, S)], 8, "START without any later END"),
    ([ev(10, 3, S), ev(20000, 3, S), ev(20100, 3, E0)], 4, "run open at the end of a chunk, next chunk opens with a START"),
    ([ev(10, 25, S), ev(60, 25, E0)], 1, "motif outside the launch"),
    ([ev(10, 3, S), ev(9 * TILE, 3, E0)], 1, "position beyond the record"),
    ([ev(10, 3, S), ev(60, 3, E0), ev(100, 5, S), ev(160, 5, E0), ev(300, 3, S), ev(360, 3, E0)], 2, "one (motif, tile) chunk in two pieces"),
])
def test_malformed_streams_raise_their_flag(events, bit, what):
    runs, flags = run(events)
    assert flags & bit, (what, flags)


def test_empty_stream():
    runs, flags = run([])
    assert flags == 0 and len(runs) == 0
