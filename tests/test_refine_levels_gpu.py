"""processSeed's recursion on the flanks (parse_seed.cpp:443-463), level by level on the GPU (refine.h: DeferredNode;
api_refine_bed.cpp: refine_levels): nodes of long-motif seeds' recursion trees from RIBBIT_DEFER_MIN bases on are not refined where
they are met but put off, batched -- consensus rows, striped passes, path search -- and their rows sorted into place.
The BED text must stay the oracle pipeline's byte for byte whatever the threshold, in both forms of refinement (a short
record's host threads, a long record's GPU alignment pipeline), and the counters must show that nodes WERE put off over
several levels."""
import os

import pytest

import ribbit_amd
from oracle_lib import Oracle
from ribbit_amd.simulate import simulate_sequence

pytestmark = pytest.mark.gpu


def _bed_both(seq, m_lo, m_hi, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        before = ribbit_amd.level_counters()
        with ribbit_amd.Scanner(m_lo, m_hi) as sc:
            sc.load_record(seq)
            got = sc.refine_bed("rec")
        after = ribbit_amd.level_counters()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    return got, tuple(b - a for a, b in zip(before, after))


@pytest.fixture(scope="module")
def long_motif_record():
    seq, _ = simulate_sequence(600_000, 31, 2, 500, n_block_rate=0.2, lower_rate=0.1)
    with Oracle(seq, 2, 500) as o:
        o.run_all()
        want = o.refine_bed("rec")
    return seq, want


@pytest.mark.parametrize("gpu_ssw", ["0", "1"], ids=["host_threads", "gpu_pipeline"])
@pytest.mark.parametrize("defer_min", ["1", "150", "700", "0"])
def test_nodes_put_off_for_the_gpu_leave_the_bed_unchanged(long_motif_record, gpu_ssw, defer_min):
    seq, want = long_motif_record
    # (RIBBIT_LEVEL_MIN=1: every level gets a GPU batch of its own, however small; by default a level of fewer than 400 nodes
    # is finished on the host threads)
    got, (levels, nodes, aligned) = _bed_both(seq, 2, 500, {"RIBBIT_GPU_SSW": gpu_ssw, "RIBBIT_DEFER_MIN": defer_min, "RIBBIT_LEVEL_MIN": "1"})
    assert got == want
    if defer_min == "0":
        assert nodes == 0 and levels == 0
    else:
        assert nodes > 0 and aligned > 0, (levels, nodes, aligned)
        if defer_min in ("1", "150"):
            assert levels >= 2, (levels, nodes, aligned)          # flanks of flanks: the tree is walked level by level


def test_every_case_with_every_node_put_off():
    """the edge cases and simulated records of the other suites with the threshold at one base: every long-motif node
    that fits a kernel goes through the level batches"""
    from cases import edge_cases, large_motif_cases, simulated_cases
    for name, seq, m_lo, m_hi in edge_cases() + simulated_cases() + large_motif_cases():
        with Oracle(seq, m_lo, m_hi) as o:
            o.run_all()
            want = o.refine_bed("rec")
        for gpu_ssw in ("0", "1"):
            for level_min in ("1", "3"):         # 3: the first levels batched, the tails by recursion on the host threads
                got, _ = _bed_both(seq, m_lo, m_hi, {"RIBBIT_GPU_SSW": gpu_ssw, "RIBBIT_DEFER_MIN": "1", "RIBBIT_LEVEL_MIN": level_min})
                assert got == want, (name, gpu_ssw, level_min)


@pytest.mark.parametrize("block", range(4))
def test_fuzzed_records_with_every_node_put_off(block):
    from fuzz import fuzz_case
    for seed in range(31000 + 25 * block, 31000 + 25 * (block + 1)):
        seq, m_lo, m_hi = fuzz_case(seed)
        with Oracle(seq, m_lo, m_hi) as o:
            o.run_all()
            want = o.refine_bed("rec")
        for gpu_ssw in ("0", "1"):
            got, _ = _bed_both(seq, m_lo, m_hi, {"RIBBIT_GPU_SSW": gpu_ssw, "RIBBIT_DEFER_MIN": "1", "RIBBIT_LEVEL_MIN": "1" if seed % 2 else "2"})
            assert got == want, (seed, gpu_ssw, len(seq), m_lo, m_hi)



def test_reads_in_flight_put_nodes_off_and_finish_them_on_their_own_threads(tmp_path):
    """ribbit-hip on many short records at -M 500: a read puts its expensive nodes off like any record, finds them too few
    for a GPU batch (5-30 a level) and finishes them on its host threads right after -- same BED as the oracle's, with the
    threshold low (most long-motif nodes take the detour) and at its default."""
    import subprocess
    from ribbit_amd.simulate import write_fasta
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seq, _ = simulate_sequence(1_200_000, 57, 2, 500)
    records = [(f"read{i}", seq[i * 60_000:(i + 1) * 60_000]) for i in range(20)]
    fa, bed = tmp_path / "in.fa", tmp_path / "out.bed"
    write_fasta(str(fa), records)
    want = []
    for name, s in records:
        with Oracle(s, 2, 500) as o:
            o.run_all()
            want.append(o.refine_bed(name))
    want = "".join(want)
    for defer_min, level_min in (("200", ""), ("", ""), ("200", "4"), ("0", "")):
        env = dict(os.environ, RIBBIT_PROFILE="1")
        if defer_min:
            env["RIBBIT_DEFER_MIN"] = defer_min
        if level_min:
            env["RIBBIT_LEVEL_MIN"] = level_min          # (levels of four nodes and more get GPU batches of their own)
        r = subprocess.run([os.path.join(root, "ribbit_amd", "ribbit-hip"), "-i", str(fa), "-o", str(bed), "-m", "2", "-M", "500", "--jobs", "6"],
                           capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        assert bed.read_text() == want, (defer_min, level_min)
