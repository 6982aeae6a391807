"""Large test records: simulated 4-Mbp segments joined by blocks of N (shared by test_whole_path_large_gpu.py).

NOT a decomposition: the BED of such a record is not the union of its segments' BEDs.  Measured with the oracle alone
(3 x 4 Mbp, round 2): the whole record has 15 rows more than the shifted union, first at a locus 33 kb into the third
segment, because the anchored merge's decisions there depend on seed-list state left by the earlier segments.  A first
version of the large test relied on that decomposition (it held at 3 x 150 kb) and "failed" against a correct GPU
result; the oracle on the whole record has the GPU's rows.  So the large test runs the oracle on the whole record."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAD = b"N" * 1000


def concatenated(segments) -> bytes:
    return PAD + PAD.join(segments) + PAD


def simulated_record(total_bases: int, first_seed: int = 500, segment_bases: int = 4_000_000) -> bytes:
    from ribbit_amd.simulate import simulate_sequence
    sizes = [segment_bases] * (total_bases // segment_bases) + ([total_bases % segment_bases] if total_bases % segment_bases else [])
    return concatenated([simulate_sequence(n, first_seed + k, 2, 100, n_block_rate=0.1, lower_rate=0.1)[0] for k, n in enumerate(sizes)])


def oracle_bed_of_simulated_record(args):
    """worker (no GPU, own process): builds simulated_record(total_bases, first_seed) itself and runs the oracle pipeline
    on the whole of it (~5 s and ~0.25 GB per Mbp on one core)"""
    total_bases, first_seed, seq_id = args
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    with Oracle(simulated_record(total_bases, first_seed), 2, 100) as o:
        o.run_all()
        return o.refine_bed(seq_id)
