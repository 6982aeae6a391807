"""End to end on the GPU: FASTA in -> BED out through the command-line front end (ribbit-hip) and
through Scanner.refine_bed, against the oracle pipeline (restated host logic around the reference's own
SSW).  BED text must be identical: integer coordinates, motif, CIGAR, and the purity column as printed."""
import os
import subprocess

import pytest

import ribbit_amd
from cases import edge_cases, large_motif_cases, simulated_cases
from oracle_lib import Oracle
from ribbit_amd.simulate import write_fasta

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "ribbit_amd", "ribbit-hip")
ALL = edge_cases() + simulated_cases() + large_motif_cases()


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
