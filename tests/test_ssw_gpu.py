"""GPU leg of the alignment row (f1): the batched striped passes of ribbit_hip_ssw_passes (one alignment per DPP row,
ssw_kernels.hip) against the REFERENCE library itself (oracle/_ref/libssw_ref.so, compiled from the reference's
vendored ssw.c / ssw_cpp.cpp) -- score, end point, second best score and begin point -- and, through
Scanner.refine_bed, the BED text of records whose alignments all went through the GPU."""
import os

import numpy as np
import pytest

import ribbit_amd
from test_ssw import REF_SO, _mutate, _rand, ref_align

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libssw_ref.so not built")]


def _batch(pairs):
    """pairs: (query, motif, ppr_len) -> record, jobs, pool: the queries become slices of one record"""
    record, pool = bytearray(), bytearray()
    jobs = np.zeros(len(pairs), ribbit_amd.JOB_DT)
    for k, (query, motif, ppr_len) in enumerate(pairs):
        jobs[k]["query_start"], jobs[k]["query_length"] = len(record), len(query)
        jobs[k]["ppr_length"], jobs[k]["atomicity"], jobs[k]["motif_offset"] = ppr_len, len(motif), len(pool)
        jobs[k]["motif_length"] = len(motif)
        record += query + b"N" * 3                       # a gap, so that an off-by-one in a slice shows
        pool += motif
    return bytes(record), jobs, bytes(pool)


def _check_batch(pairs, mask_len=15):
    record, jobs, pool = _batch(pairs)
    with ribbit_amd.Scanner(2, 8) as sc:
        sc.load_record(record)
        got = sc.ssw_passes(jobs, pool, mask_len)
    n_gpu = 0
    for k, (query, motif, ppr_len) in enumerate(pairs):
        if got[k]["flag"] == -1:
            assert len(query) > 8192 or ppr_len > 16384, (k, len(query), ppr_len)
            continue
        n_gpu += 1
        ref = motif * (ppr_len // len(motif) + 2)
        want, _ = ref_align(query, ref, ppr_len, mask_len)
        g = got[k]
        if want["sw_score"] == 0:                        # the library reads ref[-1] here (UB); defined as "no alignment"
            assert g["score"] == 0 and g["ref_end"] == -1, (k, g)
            continue
        have = (g["score"], g["ref_end"], g["query_end"], g["score2"], g["ref_end2"], g["ref_begin"], g["query_begin"])
        ref_vals = (want["sw_score"], want["ref_end"], want["query_end"], want["sw_score_next_best"], want["ref_end_next_best"],
                    want["ref_begin"], want["query_begin"])
        assert have == ref_vals, (k, query, motif, ppr_len, have, ref_vals)
    return n_gpu


@pytest.mark.parametrize("seed", range(6))
def test_repeat_like_jobs_match_reference_library(seed):
    rs = np.random.RandomState(900 + seed)
    pairs = []
    for _ in range(400):
        m = int(rs.randint(1, 30))
        motif = _rand(rs, m)
        units = int(rs.randint(2, 40))
        rot = int(rs.randint(0, m))
        pure = (motif * (units + 2))[rot:rot + m * units + int(rs.randint(0, m))]
        query = _mutate(rs, pure, float(rs.choice([0.0, 0.03, 0.1, 0.2])))
        if query:
            pairs.append((query, motif, len(query) + m + int(0.15 * len(query))))
    assert _check_batch(pairs) > 250


def test_random_pairs_and_unknown_bases_match_reference_library():
    rs = np.random.RandomState(77)
    pairs = []
    for _ in range(500):
        q = _rand(rs, int(rs.randint(1, 120)), b"ACGTN" if rs.random_sample() < 0.2 else b"ACGTacgtU")
        motif = _rand(rs, int(rs.randint(1, 200)))       # a long "motif" = an arbitrary reference
        pairs.append((q, motif, len(motif)))
    assert _check_batch(pairs) == len(pairs)


def test_long_alignments_take_the_16bit_path_and_oversized_jobs_are_left_to_the_host():
    rs = np.random.RandomState(7)
    pairs = []
    for n in (130, 140, 200, 300, 400, 500, 512, 513, 600, 1500, 2048, 2049, 2500, 4096, 4097, 5000, 8192, 8193, 9000):
        motif = _rand(rs, int(rs.randint(2, 12)))
        pure = (motif * (n // len(motif) + 2))[:n]
        query = _mutate(rs, pure, 0.05)[:n]
        pairs.append((query, motif, len(query) + len(motif) + int(0.15 * len(query))))
    n_gpu = _check_batch(pairs)
    fits = sum(1 for q, _, ppr in pairs if len(q) <= 8192 and ppr <= 16384)      # (a mutated query may be shorter than asked for)
    assert n_gpu == fits and fits < len(pairs)      # the query of 9000 bases is the host's


def test_mask_length_below_15_disables_the_second_best():
    rs = np.random.RandomState(3)
    motif = _rand(rs, 5)
    pairs = [(_mutate(rs, motif * 12, 0.1), motif, 80) for _ in range(40)]
    assert _check_batch(pairs, mask_len=7) == len(pairs)


def test_bed_is_identical_with_the_striped_passes_on_the_gpu(tmp_path):
    """RIBBIT_GPU_SSW=1 routes every first-level alignment's passes through ssw_passes_kernel; the BED text must not
    change (and equals the oracle pipeline's, which aligns with the reference library)."""
    import subprocess
    from cases import large_motif_cases, simulated_cases
    from oracle_lib import Oracle
    from ribbit_amd.simulate import write_fasta
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    records = [("sim", simulated_cases()[1][1]), ("big", large_motif_cases()[0][1])]
    fa = tmp_path / "in.fa"
    write_fasta(str(fa), records)
    want = ""
    for name, seq in records:
        with Oracle(seq, 2, 100) as o:
            o.run_all()
            want += o.refine_bed(name)
    for flag in ("1", None):
        env = dict(os.environ)
        env.pop("RIBBIT_GPU_SSW", None)
        if flag:
            env["RIBBIT_GPU_SSW"] = flag
        env["RIBBIT_PROFILE"] = "1"
        bed = tmp_path / f"out_{flag}.bed"
        r = subprocess.run([os.path.join(root, "ribbit_amd", "ribbit-hip"), "-i", str(fa), "-o", str(bed), "-m", "2", "-M", "100"],
                           capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        assert bed.read_text() == want
        assert ("with GPU passes" in r.stderr and "(0 with GPU passes" not in r.stderr.split("[refine]")[-1]) == bool(flag)
        assert ("with GPU paths" in r.stderr and ", 0 with GPU paths)" not in r.stderr.split("[refine]")[-1]) == bool(flag)


def test_a_record_whose_alignment_batches_do_not_fit_is_aligned_on_the_host(tmp_path):
    """when a batch's buffers cannot be allocated (several large records in flight on one GPU) the record's alignments run on
    the host threads instead: same BED, a note on stderr.  RIBBIT_DEBUG_FAIL_BATCHES makes the second slice fail that way,
    after the first slice's rows have already been written."""
    import subprocess
    from cases import simulated_cases
    from ribbit_amd.simulate import write_fasta
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fa = tmp_path / "in.fa"
    write_fasta(str(fa), [("sim", simulated_cases()[1][1])])
    beds = []
    for extra in ({"RIBBIT_GPU_SSW": "0"}, {"RIBBIT_GPU_SSW": "1", "RIBBIT_DEBUG_FAIL_BATCHES": "1"}):
        bed = tmp_path / f"out_{len(beds)}.bed"
        r = subprocess.run([os.path.join(root, "ribbit_amd", "ribbit-hip"), "-i", str(fa), "-o", str(bed), "-m", "2", "-M", "100"],
                           capture_output=True, text=True, timeout=900, env=dict(os.environ, **extra))
        assert r.returncode == 0, r.stderr[-2000:]
        assert ("GPU alignment batches skipped" in r.stderr) == ("RIBBIT_DEBUG_FAIL_BATCHES" in extra)
        beds.append(bed.read_text())
    assert beds[0] == beds[1] and beds[0].count("\n") > 100


def _check_whole_alignments(pairs, need_paths):
    """passes + banded path search on the GPU (ssw_kernels.hip, ssw_path.hip), CIGAR text on the host: everything
    Aligner::Align returns must equal the reference library's"""
    record, jobs, pool = _batch(pairs)
    with ribbit_amd.Scanner(2, 8) as sc:
        sc.load_record(record)
        got, on_gpu = sc.ssw_align_jobs(jobs, pool)
    for k, (query, motif, ppr_len) in enumerate(pairs):
        ref = motif * (ppr_len // len(motif) + 2)
        want, want_cigar = ref_align(query, ref, ppr_len)
        res, cigar = got[k]
        if want["sw_score"] == 0:                        # the library reads ref[-1] here (UB); defined as one soft clip
            assert cigar == f"{len(query)}S", (k, cigar)
            continue
        assert cigar == want_cigar, (k, query, motif, ppr_len, cigar, want_cigar, int(on_gpu[k]))
        for f in ("sw_score", "ref_begin", "ref_end", "query_begin", "query_end", "mismatches", "flag"):
            assert res[f] == want[f], (k, f, res, want)
    assert int((on_gpu == 2).sum()) >= need_paths, on_gpu
    return on_gpu


@pytest.mark.parametrize("seed", range(4))
def test_gpu_path_search_gives_the_reference_librarys_cigars(seed):
    rs = np.random.RandomState(1900 + seed)
    pairs = []
    for _ in range(600):
        m = int(rs.randint(1, 30))
        motif = _rand(rs, m)
        units = int(rs.randint(2, 40))
        rot = int(rs.randint(0, m))
        pure = (motif * (units + 2))[rot:rot + m * units + int(rs.randint(0, m))]
        query = _mutate(rs, pure, float(rs.choice([0.0, 0.03, 0.1, 0.2, 0.35])))
        if query:
            pairs.append((query, motif, len(query) + m + int(0.15 * len(query))))
    _check_whole_alignments(pairs, need_paths=400)


def test_gpu_path_search_with_wide_bands_and_long_queries():
    """indel-rich long alignments: the band doubles several times, rows wider than a wavefront are walked in chunks"""
    rs = np.random.RandomState(31)
    pairs = []
    for n in (90, 150, 260, 400, 480, 500, 700, 1200, 1900):
        for rate in (0.05, 0.15, 0.3):
            motif = _rand(rs, int(rs.randint(2, 14)))
            pure = (motif * (n // len(motif) + 2))[:n]
            query = _mutate(rs, pure, rate)[:n]
            # long deletions / insertions move the path far off the diagonal
            cut = int(rs.randint(10, 60))
            query = query[:len(query) // 2] + query[len(query) // 2 + cut:] if rs.random_sample() < 0.5 else query[:len(query) // 3] + _rand(rs, cut) + query[len(query) // 3:]
            query = query[:2048]
            pairs.append((query, motif, min(4096, len(query) + len(motif) + int(0.15 * len(query)))))
    _check_whole_alignments(pairs, need_paths=20)


def test_gpu_path_search_on_random_pairs_and_unknown_bases():
    rs = np.random.RandomState(78)
    pairs = []
    for _ in range(500):
        q = _rand(rs, int(rs.randint(1, 120)), b"ACGTN" if rs.random_sample() < 0.2 else b"ACGTacgtU")
        motif = _rand(rs, int(rs.randint(1, 200)))
        pairs.append((q, motif, len(motif)))
    _check_whole_alignments(pairs, need_paths=300)


def test_wave_kernel_on_long_queries_of_every_kind():
    """ssw_wave.hip (queries of 129..2048 bases, one wavefront per alignment): low-scoring pairs whose 8-bit pass runs to the
    end in its wave form, heavily mutated repeats around the 255 overflow threshold, ragged stripe counts (lengths around the
    multiples of 8, 16 and 64), unknown bases, references shorter and longer than the query -- passes and whole alignments
    equal the reference library's"""
    rs = np.random.RandomState(4242)
    pairs = []
    for n in list(range(129, 150)) + [159, 160, 161, 191, 192, 193, 255, 256, 257, 511, 512, 513, 640, 1023, 1024, 1025, 2047, 2048, 2049, 3000, 4095, 4096]:
        motif = _rand(rs, int(rs.randint(1, 40)))
        pure = (motif * (n // len(motif) + 2))[:n]
        for rate in (0.0, 0.2, 0.45):
            q = _mutate(rs, pure, rate)[:n]
            if len(q) > 128:
                pairs.append((q, motif, min(8192, len(q) + len(motif) + int(0.15 * len(q)))))
    for _ in range(150):                                   # unrelated query and reference: scores of a few dozen, many lazy-F rounds
        q = _rand(rs, int(rs.randint(129, 700)), b"ACGTN" if rs.random_sample() < 0.3 else b"ACGT")
        ref = _rand(rs, int(rs.randint(20, 900)))
        pairs.append((q, ref, len(ref)))
    for _ in range(60):                                    # a short repeat inside a long unrelated query, reference far shorter / longer
        motif = _rand(rs, int(rs.randint(2, 9)))
        core = _mutate(rs, motif * int(rs.randint(5, 60)), 0.1)
        q = _rand(rs, int(rs.randint(60, 300))) + core + _rand(rs, int(rs.randint(60, 300)))
        if 128 < len(q) <= 2048:
            pairs.append((q, motif, int(rs.choice([40, len(q) // 2, len(q) + 50, 2 * len(q)]))))
    assert _check_batch(pairs) == len(pairs)
    _check_whole_alignments(pairs, need_paths=len(pairs) // 2)


def test_group_kernel_on_long_queries_of_every_kind():
    """ssw_group.hip (queries of 513..8192 bases, a workgroup of 4, 8 or 16 wavefronts per alignment, a column's stripes dealt to
    the wavefronts): what is new there is the carry between wavefronts, the lazy-F loop of the first wavefront running on into
    stripes the others own, and stripe counts that do not divide by the wavefronts -- so: unrelated long pairs (scores of a few
    dozen: F is raised again and again), long gaps (a block deleted from / inserted into a repeat: F carries across many
    stripes), unknown bases, lengths around every multiple that matters, references far shorter and far longer than the query.
    Passes and whole alignments equal the reference library's."""
    rs = np.random.RandomState(777)
    pairs = []
    lengths = [513, 514, 519, 520, 521, 527, 528, 529, 543, 544, 545, 575, 576, 577, 767, 768, 769, 1000, 1031, 1279, 1280, 1281, 1536, 1537,
               2047, 2048, 2049, 2050, 2063, 2064, 2065, 2111, 2112, 2113, 2560, 3071, 3072, 3073, 3583, 3584, 3585, 4000, 4031, 4032, 4033, 4080, 4095, 4096,
               4097, 4100, 4111, 4112, 4113, 4608, 5000, 6143, 6144, 6145, 7000, 8063, 8064, 8065, 8190, 8191, 8192]      # 16 wavefronts, 124 KB of LDS
    for n in lengths:
        motif = _rand(rs, int(rs.randint(1, 60)))
        pure = (motif * (n // len(motif) + 2))[:n]
        q = _mutate(rs, pure, float(rs.choice([0.0, 0.1, 0.3, 0.45])))[:n]
        if len(q) > 512:
            pairs.append((q, motif, min(16384, len(q) + len(motif) + int(0.15 * len(q)))))
    for n in (600, 900, 1500, 2048, 2100, 3000, 4096, 5001, 8192):        # unrelated: many lazy-F rounds, on all three widths
        q = _rand(rs, n, b"ACGTN" if n % 2 else b"ACGT")
        ref = _rand(rs, int(rs.randint(50, 1200)))
        pairs.append((q, ref, len(ref)))
        pairs.append((q, _rand(rs, 7), min(16384, n + 200)))
    for n in (700, 1800, 2500, 3900, 6000, 8000):              # a long gap either way inside a clean repeat
        motif = _rand(rs, int(rs.randint(3, 30)))
        pure = (motif * (n // len(motif) + 2))[:n]
        cut = int(rs.randint(100, n - 300))
        gap = int(rs.randint(20, 150))
        pairs.append((pure[:cut] + pure[cut + gap:], motif, min(16384, n + 100)))                   # deletion from the query
        pairs.append(((pure[:cut] + _rand(rs, gap) + pure[cut:])[:8192], motif, min(16384, n + 100)))  # insertion into it
    for n in (1000, 2600, 4096, 8192):                         # reference far shorter than the query
        motif = _rand(rs, 5)
        pairs.append((_mutate(rs, (motif * (n // 5 + 2))[:n], 0.05)[:n], motif, 60))
    assert _check_batch(pairs) == len(pairs)
    _check_whole_alignments(pairs, need_paths=1)


def test_two_batches_of_different_size_on_one_handle():
    """Regression for a crash of round 2 (feeder thread of ribbit_hip_refine_bed): the path search's result slots are reused
    from batch to batch, and a second, smaller batch read a path-operation count the first batch had left in a slot it did not
    write itself (jobs without a path: score 0, or beyond the distance filter).  One handle, a large batch, then a smaller one
    in another order that contains such jobs: every record of both must equal the reference library's."""
    rs = np.random.RandomState(20509)
    big = []
    for _ in range(700):
        m = int(rs.randint(1, 30))
        motif = _rand(rs, m)
        pure = (motif * 42)[int(rs.randint(0, m)):][:m * int(rs.randint(2, 40))]
        query = _mutate(rs, pure, float(rs.choice([0.0, 0.05, 0.2])))
        if query:
            big.append((query, motif, len(query) + m + int(0.15 * len(query))))
    small = [big[k] for k in rs.permutation(len(big))[:90]]
    # jobs that end without a path: nothing in common with the reference (score 0 -> one soft clip)
    for _ in range(25):
        small.insert(int(rs.randint(0, len(small))), (b"A" * int(rs.randint(1, 60)), b"C" * int(rs.randint(1, 9)), 40))
    with ribbit_amd.Scanner(2, 8) as sc:
        for pairs in (big, small):
            record, jobs, pool = _batch(pairs)
            sc.load_record(record)
            got, on_gpu = sc.ssw_align_jobs(jobs, pool)
            for k, (query, motif, ppr_len) in enumerate(pairs):
                ref = motif * (ppr_len // len(motif) + 2)
                want, want_cigar = ref_align(query, ref, ppr_len)
                res, cigar = got[k]
                if want["sw_score"] == 0:
                    assert cigar == f"{len(query)}S", (k, cigar)
                    continue
                assert cigar == want_cigar, (k, cigar, want_cigar, int(on_gpu[k]))
                for f in ("sw_score", "ref_begin", "ref_end", "query_begin", "query_end", "mismatches", "flag"):
                    assert res[f] == want[f], (k, f, res, want)
            assert int((on_gpu == 2).sum()) >= len(pairs) // 2
