"""End to end on the GPU: FASTA in -> BED out through the command-line front end (ribbit-hip) and
through Scanner.refine_bed, against the oracle pipeline (restated host logic around the reference's own
SSW).  BED text must be identical: integer coordinates, motif, CIGAR, and the purity column as printed."""
import os
import subprocess

import pytest

import ribbit_amd
from cases import edge_cases, large_motif_cases, simulated_cases, structured_cases
from oracle_lib import Oracle
from ribbit_amd.simulate import write_fasta

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "ribbit_amd", "ribbit-hip")
ALL = edge_cases() + simulated_cases() + large_motif_cases() + structured_cases()


@pytest.mark.parametrize("name,seq,m_lo,m_hi", ALL, ids=[c[0] for c in ALL])
def test_refine_bed_matches_oracle(name, seq, m_lo, m_hi):
    with ribbit_amd.Scanner(m_lo, m_hi) as sc, Oracle(seq, m_lo, m_hi) as o:
        sc.load_record(seq)
        o.run_all()
        assert sc.refine_bed(name).split("\n") == o.refine_bed(name).split("\n")


def _oracle_bed(records, m_lo, m_hi):
    out = []
    for rec_name, seq in records:
        with Oracle(seq, m_lo, m_hi) as o:
            o.run_all()
            out.append(o.refine_bed(rec_name.split(" ")[0]))
    return "".join(out)


def test_cli_multi_record_fasta(tmp_path):
    sims = simulated_cases()
    records = [("chrA first record", sims[0][1][:60_000]), ("chrB", sims[1][1][:50_000]), ("chrC_with_N", edge_cases()[10][1] * 20)]
    fa, bed = tmp_path / "in.fa", tmp_path / "out.bed"
    write_fasta(str(fa), records)
    r = subprocess.run([BIN, "-i", str(fa), "-o", str(bed), "-m", "2", "-M", "30"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Processing sequence chrA" in r.stderr and "Total number of perfect seeds:" in r.stderr
    assert bed.read_text() == _oracle_bed(records, 2, 30)


def test_cli_records_dealt_over_several_devices(tmp_path):
    """ribbit.cpp:269-280 handles the records one after the other; --devices deals them over GPUs (here GPU 0 listed
    twice: two handle sets, the longest record of the look-ahead first) and the BED keeps the input order."""
    sims = simulated_cases()
    records = [(f"rec{i}", sims[i % 2][1][i * 7000:i * 7000 + 9000 + 4000 * (i % 5)]) for i in range(12)]
    records.insert(3, ("long one", sims[1][1][:150_000]))
    fa, bed, bed2 = tmp_path / "in.fa", tmp_path / "out.bed", tmp_path / "out2.bed"
    write_fasta(str(fa), records)
    env = dict(os.environ, RIBBIT_PROFILE="1")
    r = subprocess.run([BIN, "-i", str(fa), "-o", str(bed), "-m", "2", "-M", "30", "--devices", "0,0", "--jobs", "2"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    want = _oracle_bed(records, 2, 30)
    assert bed.read_text() == want
    dealt = [l for l in r.stderr.split("\n") if l.startswith("[devices] slot")]
    assert len(dealt) == 2 and all(" 0 records" not in l for l in dealt), r.stderr[-1500:]
    # ... and on one slot with three records in flight the BED is the same
    r = subprocess.run([BIN, "-i", str(fa), "-o", str(bed2), "-m", "2", "-M", "30", "--jobs", "3"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, RIBBIT_PROFILE="1"))
    assert r.returncode == 0 and bed2.read_text() == want
    # the same list through the environment; an unusable list is refused
    r = subprocess.run([BIN, "-i", str(fa), "-o", str(bed2), "-m", "2", "-M", "30"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, RIBBIT_DEVICES="0,0,0"))
    assert r.returncode == 0 and bed2.read_text() == want
    r = subprocess.run([BIN, "-i", str(fa), "--devices", "0,x"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "--devices" in r.stderr


@pytest.mark.parametrize("label,m_lo,m_hi", [("edge_cases", 2, 12), ("fuzz_batch", 2, 40)])
def test_cli_with_every_alignment_forced_onto_the_gpu(tmp_path, label, m_lo, m_hi):
    """RIBBIT_GPU_SSW=1: the alignments of EVERY record go through the GPU batches (by default only records with 400,000
    dispatched seeds or more do).  Edge cases (empty-ish records, N blocks, runs at both record ends) and one batch of
    fuzz records, BED against the oracle pipeline."""
    if label == "edge_cases":
        records = [(n, s) for n, s, _, _ in edge_cases() if len(s) > 0]
    else:
        from fuzz import fuzz_case
        records = [(f"fz{seed}", fuzz_case(seed)[0]) for seed in range(31500, 31530)]
        records = [(n, s.replace(b"\n", b"")) for n, s in records if len(s) > 0]
    fa, bed = tmp_path / "in.fa", tmp_path / "out.bed"
    write_fasta(str(fa), records)
    r = subprocess.run([BIN, "-i", str(fa), "-o", str(bed), "-m", str(m_lo), "-M", str(m_hi)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, RIBBIT_GPU_SSW="1", RIBBIT_PROFILE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert bed.read_text() == _oracle_bed(records, m_lo, m_hi)
    assert "alignment jobs (" in r.stderr, "no record took the GPU alignment path"


def test_the_alignment_pipeline_on_every_size_class_and_any_number_of_slices(tmp_path):
    """One record with long repeats -- alignments of every size class incl. the three workgroup classes -- with the alignments
    forced onto the GPU pipeline, through the command line, cut into 2, 5 and 9 slices of the seed list (two feeders on
    alternating slices): the BED is the oracle's every time.  (Until round 4 this test also ran the pipeline's older forms
    behind RIBBIT_SSW_FEEDERS / RIBBIT_SSW_GROUP; they lost their measurements in round 3 and are gone from the product.)"""
    import numpy as np
    from ribbit_amd.simulate import simulate_sequence
    seq, _ = simulate_sequence(1_500_000, 23, 2, 60)
    rs = np.random.RandomState(5)
    long_repeats = b""
    for unit_len, copies in ((7, 300), (23, 180), (41, 150), (13, 500)):        # queries of 2..6.5 kb: the three workgroup classes
        unit = bytes(rs.choice(list(b"ACGT"), unit_len).astype(np.uint8))
        body = bytearray(unit * copies)
        for k in rs.choice(len(body), len(body) // 40, replace=False):          # a mutation every 40 bases
            body[k] = b"ACGT"[(b"ACGT".index(body[k]) + 1 + int(rs.randint(3))) % 4]
        long_repeats += bytes(rs.choice(list(b"ACGT"), 500).astype(np.uint8)) + bytes(body)
    records = [("long_repeats", seq[:700_000] + long_repeats + seq[700_000:])]
    fa = tmp_path / "in.fa"
    write_fasta(str(fa), records)
    beds = {}
    for slices in ("2", "5", "9"):
        bed = tmp_path / f"slices{slices}.bed"
        r = subprocess.run([BIN, "-i", str(fa), "-o", str(bed), "-m", "2", "-M", "60"], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, RIBBIT_GPU_SSW="1", RIBBIT_SSW_SLICES=slices, RIBBIT_PROFILE="1"))
        assert r.returncode == 0, r.stderr[-2000:]
        assert "alignment jobs (" in r.stderr and f"{slices} slices" in r.stderr, r.stderr[-1500:]
        beds["default" if slices == "5" else slices] = bed.read_text()
    assert beds["default"] == _oracle_bed(records, 2, 60)
    assert all(b == beds["default"] for b in beds.values()), [k for k, b in beds.items() if b != beds["default"]]


def test_cli_long_reads_at_M_500(tmp_path):
    """BASELINE.json configs[4]: many short records (simulated long reads), -m 2 -M 500."""
    big = large_motif_cases()[0][1]
    sim = simulated_cases()[1][1]
    records = [(f"read{i}", (big[i * 900:i * 900 + 5000] + sim[i * 3000:i * 3000 + 4000])) for i in range(4)]
    fa, bed = tmp_path / "in.fa", tmp_path / "out.bed"
    write_fasta(str(fa), records)
    r = subprocess.run([BIN, "-i", str(fa), "-o", str(bed), "-m", "2", "-M", "500"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert bed.read_text() == _oracle_bed(records, 2, 500)


def test_cli_bed_goes_to_stderr_without_o_and_purity_flag_is_ignored(tmp_path):
    rec = [("r1", simulated_cases()[0][1][:30_000])]
    fa = tmp_path / "in.fa"
    write_fasta(str(fa), rec)
    r = subprocess.run([BIN, "-i", str(fa), "-m", "2", "-M", "6", "-p", "0.5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout == ""
    want = _oracle_bed(rec, 2, 6)
    got = "".join(l + "\n" for l in r.stderr.split("\n") if l.startswith("r1\t"))
    assert got == want and "Purity threshold: 0.85" in r.stderr


def test_cli_help_exits_1():
    r = subprocess.run([BIN, "--help"], capture_output=True, text=True)
    assert r.returncode == 1 and "--min-motif-length" in r.stderr
    r = subprocess.run([BIN], capture_output=True, text=True)
    assert r.returncode == 1 and "Please specify an input fasta file" in r.stderr


def _reference_tables(m_lo, m_hi, min_length=None, min_units=None, perfect_units=None):
    """MINIMUM_LENGTH / PERFECT_UNITS as ribbit.cpp:143-174 and :210-235 fill them.  An option value is an int
    ("same for every motif size in range") or a dict read from a two-column file."""
    def dual(value):
        return {k: value for k in range(m_lo, m_hi + 1)} if isinstance(value, int) else dict(value)
    ml, pu = {}, {}
    if min_length is not None:
        ml = dual(min_length)
    elif min_units is not None:
        ml = {k: k * v for k, v in dual(min_units).items()}
    else:
        ml = {k: max(12, 2 * k) for k in range(m_lo, m_hi + 1)}
    if perfect_units is not None:
        pu = dual(perfect_units)
    else:
        pu = {m: {1: 8, 2: 4, 3: 3}.get(m, 2) for m in range(1, m_hi + 1)}
    for m in range(m_lo, m_hi + 1):
        for f in range(1, m // 2 + 1):
            if m % f:
                continue
            if f not in ml:
                ml[f] = ml.setdefault(m, 0)          # operator[] inserts 0 for a missing key
            if f not in pu:
                pu[f] = pu.setdefault(m, 0) * (m // f)
    return ml, pu


def _oracle_params(m_lo, m_hi, ml, pu):
    from oracle_lib import RefineParams as OracleRefineParams
    import ctypes as C
    from oracle_lib import lib
    rp = OracleRefineParams()
    lib().rbo_refine_params_default(C.byref(rp), m_lo, m_hi)
    for k in range(1024):
        rp.min_length[k] = ml.get(k, 0)
        rp.perfect_units[k] = pu.get(k, 0)
    return rp


@pytest.mark.parametrize("label,m_lo,m_hi,flags,tables", [
    ("min_length_number", 2, 12, ["-l", "20"], dict(min_length=20)),
    ("min_units_number", 2, 12, ["--min-units", "4"], dict(min_units=4)),
    ("perfect_units_number", 3, 12, ["--perfect-units", "3"], dict(perfect_units=3)),
    ("both_numbers", 4, 30, ["--min-units=3", "--perfect-units", "2"], dict(min_units=3, perfect_units=2)),
    ("files", 2, 10, None, dict(min_units={2: 8, 3: 5, 4: 4, 5: 4, 6: 3, 8: 3, 10: 2}, perfect_units={2: 5, 3: 4, 4: 3, 5: 3, 6: 2, 7: 2, 9: 2})),
])
def test_cli_length_and_unit_options_follow_the_reference_tables(tmp_path, label, m_lo, m_hi, flags, tables):
    """-l / --min-units / --perfect-units as numbers and as two-column files (ribbit.cpp:25-64,143-174), incl. the
    completion of the tables for factors of the selected motif sizes (:210-235): BED against the oracle pipeline
    run with the tables a restatement of those lines produces."""
    seq = simulated_cases()[2][1] + simulated_cases()[0][1][:40_000]
    fa, bed = tmp_path / "in.fa", tmp_path / "out.bed"
    write_fasta(str(fa), [("rec", seq)])
    if flags is None:
        flags = []
        for opt, key in (("--min-units", "min_units"), ("--perfect-units", "perfect_units")):
            path = tmp_path / f"{key}.tsv"
            path.write_text("".join(f"{k}\t{v}\n" for k, v in tables[key].items()))
            flags += [opt, str(path)]
    r = subprocess.run([BIN, "-i", str(fa), "-o", str(bed), "-m", str(m_lo), "-M", str(m_hi)] + flags, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    ml, pu = _reference_tables(m_lo, m_hi, **tables)
    with Oracle(seq, m_lo, m_hi) as o:
        o.run_all()
        want = o.refine_bed("rec", _oracle_params(m_lo, m_hi, ml, pu))
    assert bed.read_text() == want
    assert want.count("\n") > 20


def test_cli_reads_fasta_the_way_getline_does(tmp_path):
    """ribbit.cpp:269-280: every line that does not start with '>' is appended as it is -- blank lines add nothing,
    a carriage return (Windows line ends) becomes a base that encodes as N, the name is the header up to the first
    space (the whole rest of the line, carriage return included, when there is none)."""
    sims = simulated_cases()
    a, b, c = sims[0][1][:30_000], sims[2][1][:30_000], sims[1][1][:20_000]

    def wrap(seq, width, eol):
        return b"".join(seq[i:i + width] + eol for i in range(0, len(seq), width))

    text = (b">chrA description with spaces\n" + wrap(a, 60, b"\n") + b"\n\n"
            + b">chrB\r\n" + wrap(b, 70, b"\r\n")
            + b">chrC\tTabbed\n" + wrap(c, 80, b"\n") + c[:10])            # no newline at the end of the file
    fa, bed = tmp_path / "in.fa", tmp_path / "out.bed"
    fa.write_bytes(text)
    r = subprocess.run([BIN, "-i", str(fa), "-o", str(bed), "-m", "2", "-M", "12"], capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    # what the reader hands to processSequence
    records, name, seq = [], None, b""
    for line in text.split(b"\n"):
        if line[:1] == b">":
            if seq:
                records.append((name, seq))
            sp = line.find(b" ")
            name, seq = (line[1:] if sp < 0 else line[1:sp]), b""
        else:
            seq += line
    records.append((name, seq))
    assert [n for n, _ in records] == [b"chrA", b"chrB\r", b"chrC\tTabbed"] and b"\r" in records[1][1]
    want = b""
    for n, s in records:
        with Oracle(s, 2, 12) as o:
            o.run_all()
            want += o.refine_bed(n.decode()).encode()
    assert bed.read_bytes() == want and want.count(b"\n") > 100
    assert r.stderr.count(b"Processing sequence") == 2                       # not for the last record (:280)


def test_whole_path_on_a_two_megabase_record_matches_oracle():
    """Scale check of the whole path (scans, compact anchored calls, merges, refinement on worker threads, BED) on
    a record of ~125 kernel tiles with N blocks and lower case, against the oracle pipeline."""
    from ribbit_amd.simulate import simulate_sequence
    seq, _ = simulate_sequence(2_000_000, 97, 2, 100, n_block_rate=0.2, lower_rate=0.2)
    with ribbit_amd.Scanner(2, 100) as sc, Oracle(seq, 2, 100) as o:
        sc.load_record(seq)
        o.run_all()
        got = sc.refine_bed("chrTest")
        want = o.refine_bed("chrTest")
    assert got == want and want.count("\n") > 15_000


def test_cli_one_record_refined_over_several_devices(tmp_path):
    """`--devices` with ONE long record: scans and merges on the first GPU, the dispatched seeds refined in slices over all the
    listed GPUs (here GPU 0 three times: three handles, three slices; RIBBIT_SHARD_MIN_SEEDS lowers the size from which that is
    done), BED identical to the oracle's and to the one-device run."""
    seq = simulated_cases()[1][1]
    records = [("chrLong some description", seq[:110_000])]
    fa, bed, bed1 = tmp_path / "in.fa", tmp_path / "out.bed", tmp_path / "one.bed"
    write_fasta(str(fa), records)
    want = _oracle_bed(records, 2, 100)
    r = subprocess.run([BIN, "-i", str(fa), "-o", str(bed), "-m", "2", "-M", "100", "--devices", "0,0,0"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, RIBBIT_PROFILE="1", RIBBIT_SHARD_MIN_SEEDS="500"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert bed.read_text() == want
    assert "in 3 slices over as many handles" in r.stderr, r.stderr[-1500:]
    timing = tmp_path / "timing.json"
    r = subprocess.run([BIN, "-i", str(fa), "-o", str(bed1), "-m", "2", "-M", "100", "--timing", str(timing)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and bed1.read_text() == want
    # --timing: a JSON record of the run (the reference has progress lines on stderr only)
    import json
    t = json.loads(timing.read_text())
    assert t["records"] == 1 and t["bases"] == 110_000 and t["status"] == 0 and t["wall_s"] > 0
    assert set(t["stage_ms_summed_over_records"]) == {"load", "perfect", "substitutions", "anchored", "dispatch", "refine_and_bed"}
    assert t["stage_ms_summed_over_records"]["refine_and_bed"] > 0
    # several records: only the last one (processed when the others are done and the GPUs idle) is dealt that way
    records = [("first", seq[:40_000]), ("second", seq[40_000:90_000]), ("third and last", seq[90_000:200_000])]
    write_fasta(str(fa), records)
    r = subprocess.run([BIN, "-i", str(fa), "-o", str(bed), "-m", "2", "-M", "100", "--devices", "0,0"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, RIBBIT_PROFILE="1", RIBBIT_SHARD_MIN_SEEDS="500"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert bed.read_text() == _oracle_bed(records, 2, 100)
    assert r.stderr.count("slices over as many handles") == 1 and "refinement of third" in r.stderr, r.stderr[-1500:]
