#!/usr/bin/env python3
"""bench.py -- Gbases/s of the shift-XOR scan (BASELINE.json metric) on N MI355X GPUs.

One "step" = one pass of the hot path over one record that is already resident in HBM as ASCII:
pack kernel (fasta_utils.cpp:78-115) + perfect shift-XOR scan kernel over m=2..100
(fasta_utils.cpp:117-122 + parse_perfect_shiftxor.cpp:173-223) + the pairing kernels that turn its START / END
events into run records on the device + one D2H copy of those records (16 B each).  Workload = BASELINE.json configs[1]: 100 Mbp synthetic FASTA, -m 2 -M 100, perfect scan.

Next to that line's headline (configs[1]) rank 0 of an N = 1 run also reports, in the same JSON object:
  verified          the GPU's perfect-stage calls on the first 20 Mbp equal the CPU oracle's (BASELINE: "BED diff==0")
  pcie_inclusive    the same step with the bases starting in page-locked HOST memory (clock from the first H2D enqueue,
                    uploads double-buffered against the previous batch's kernels): SURVEY.md 8(d)'s clock
  pack_hbm          pack_kernel over a rotating working set larger than the 256 MiB Infinity Cache
  full_path_sample  10 Mbp of the workload through the whole path, beside the CPU oracle on 1 Mbp
  m500_full_path    BASELINE.json configs[4]'s motif range (-m 2 -M 500) on a 16-Mbp record, FASTA record -> BED text, checked against
                    the oracle pipeline's committed digest
  chr1_full_path    BASELINE.json configs[2]'s largest record: one chromosome-1-sized record (248,956,422 bp, generator
                    seed 4, N blocks) through perfect + substitution + anchored scans, merges, dispatch, refinement
                    and BED text, with per-kernel times and a roofline block for each scan kernel
  cpu_baseline      the oracle's perfect stage on 20 Mbp, one core (CPU model and core count stated)
and an N > 1 run reports
  full_path_sharded one record of N x 10 Mbp chunk-sharded over the ranks through perfect + substitution + anchored
                    stages: device stages per rank, 16 bytes per kept call to rank 0, merges there; checked against rank
                    0's own scan of the whole record; bytes every rank sent

N > 1: one process per GPU (torch.distributed, backend nccl == RCCL).  `python bench.py --gpus N` without a launcher
starts `python -m torch.distributed.run --nproc-per-node N` itself (before anything touches a GPU) and exits with its
code; under a launcher WORLD_SIZE must equal --gpus.  ONE record of N x 100 Mbp is
chunk-sharded (option 2): every rank owns a 100-Mbp chunk (weak scaling), scans it
together with halos of a few hundred bases taken from its neighbours, keeps the events it owns, and the
chunk's events are paired locally; the sparse run records (candidate seed intervals) and the few runs
that cross a chunk edge are gathered over RCCL/xGMI before the host-side merge -- what BASELINE.json's
north_star prescribes.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOAD_BASES = 100_000_000
M_LO, M_HI = 2, 100
ALGO_BYTES_PER_BASE = 0.375      # 2 code bits + 1 N bit per base read by the scan kernel
def alignbit_share():
    """v_alignbit_b32 among the executed VALU instructions of scan_perfect_kernel: profiles/isa_mix.json (measured dynamic
    weights of the kernel's parts x exact content of its branch-free ISA blocks, tools/isa_mix.py) -> (share, bracket, source)"""
    path = os.path.join(ROOT, "profiles", "isa_mix.json")
    d = json.load(open(path))
    return float(d["alignbit_share"]), [float(x) for x in d["alignbit_share_bracket"]], "profiles/isa_mix.json"


ALIGNBIT_RATE = 550e9            # wave-instr/s chip-wide at 4 waves/SIMD (profiles/r01b_valu_peak_probe.txt)
PLAIN_VALU_RATE = 930e9          # v_or / v_xor / v_bitop3, same probe
CPU_FULL_PATH_BASES = 1_000_000  # whole-path CPU oracle sample (a few seconds)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s
CPU_SAMPLE_BASES = 20_000_000
CHR1_BASES = 248_956_422         # GRCh38 chromosome 1 (the largest record of configs[2])


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seq: bytes):
    """Oracle ("port") timed on this box's host cores, single thread, on a bounded sample.
    -> (record, the oracle's perfect-stage call list and seed list of that sample, for the verification leg)"""
    from oracle_lib import LIST_PERFECT, Oracle
    sample = seq[:CPU_SAMPLE_BASES]
    t0 = time.perf_counter()
    with Oracle(sample, M_LO, M_HI) as o:
        o.run_perfect()
        dt = time.perf_counter() - t0
        calls, seeds = o.calls(LIST_PERFECT), o.seeds(LIST_PERFECT)
    rec = {"value": len(sample) / dt / 1e9, "unit": "Gbases/s", "cores": 1, "kind": "port",
           "host_cpu": cpu_model(), "host_cores_available": os.cpu_count(),
           "sample": f"first {len(sample)} bases of the workload, encode + sweep + perfect scan + addSeed, m={M_LO}..{M_HI}, {dt:.1f} s"}
    return rec, calls, seeds


def verify_against_oracle(sc, seq: bytes, calls, seeds):
    """BASELINE.json's metric carries "BED diff==0": the GPU path's perfect-stage output on the CPU baseline's sample must
    be the oracle's, call for call and seed for seed (integer coordinates, bit-exact)."""
    import numpy as np
    sc.load_record(seq[:CPU_SAMPLE_BASES])
    got_calls = sc.perfect_calls()
    got_seeds = sc.processShiftXORsPerfect()
    ok = bool(np.array_equal(got_calls.view("<i4"), calls.view("<i4")) and np.array_equal(got_seeds.view("<i4"), seeds.view("<i4")))
    return ok, {"bases": min(len(seq), CPU_SAMPLE_BASES), "perfect_calls": int(len(got_calls)), "perfect_seeds": int(len(got_seeds)),
                "what": "GPU perfect-stage call list and seed list == CPU oracle's on the cpu_baseline sample (bit-exact)"}


def pcie_inclusive(ribbit_amd, seq: bytes, steps: int, depth: int, device: int):
    """SURVEY.md 8(d)'s clock: the step with the bases starting in page-locked host memory.  Timed from the first H2D
    enqueue to the last batch's run records on the host; every handle has its own streams, so batch k+1's upload
    (1 B/base over PCIe) overlaps batch k's kernels and batch k-1's result copy."""
    import numpy as np
    n = len(seq)
    bufs = [ribbit_amd.PinnedBuffer(n) for _ in range(depth)]
    for b in bufs:
        b.array[:] = np.frombuffer(seq, dtype=np.uint8)
    scs = [ribbit_amd.Scanner(M_LO, M_HI, device=device) for _ in range(depth)]
    for h in scs:
        h.set_timing(False)

    def issue(k):
        h = scs[k % depth]
        h.load_record_pinned(bufs[k % depth].ptr, n)
        h.scan_perfect_begin(0, (1 << 63) - 1, 0)

    def run(nsteps):
        issued, runs = 0, 0
        for k in range(min(depth - 1, nsteps)):
            issue(issued); issued += 1
        for k in range(nsteps):
            if issued <= k:
                issue(issued); issued += 1
            h = scs[k % depth]
            out = h.scan_perfect_end(wait=False)
            if issued < nsteps:
                issue(issued); issued += 1
            h.scan_perfect_wait()
            runs = len(out[0]) if isinstance(out, tuple) else len(out)
        return runs

    run(2)
    t0 = time.perf_counter()
    runs = run(steps)
    dt = time.perf_counter() - t0
    for h in scs:
        h.close()
    for b in bufs:
        b.close()
    return {"value": n * steps / dt / 1e9, "unit": "Gbases/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "batches_in_flight": depth,
            "h2d_bytes_per_step": n, "h2d_gb_per_s_needed": n / (dt / steps) / 1e9, "runs_per_step": int(runs),
            "what": "pack + perfect scan + pairing + D2H of the runs with the ASCII bases starting in page-locked host memory; clock from "
                    "the first H2D enqueue (ribbit_hip_load_record_pinned: async upload on the handle's upload stream)"}


def pack_hbm(ribbit_amd, torch, seq: bytes, dev, device: int):
    """pack_kernel over a rotating working set that cannot sit in the 256 MiB Infinity Cache: 8 distinct 100-MB ASCII
    buffers (plus 3 x 12.5 MB of planes written per launch), HIP-event time per launch."""
    import numpy as np
    copies = 8
    base = torch.frombuffer(bytearray(seq), dtype=torch.uint8).to(dev)
    bufs = [base] + [base.clone() for _ in range(copies - 1)]
    torch.cuda.synchronize()
    ms = []
    with ribbit_amd.Scanner(M_LO, M_HI, device=device) as sc:
        for rep in range(3 * copies):
            sc.load_record_device(bufs[rep % copies].data_ptr(), len(seq))
            if rep >= copies:
                ms.append(sc.timing_ms(0))
    t = float(np.median(ms))
    nbytes = len(seq) * (1 + 3 / 8)
    return {"kernel": "pack_kernel", "ms": t, "bytes_per_launch": nbytes, "achieved": nbytes / (t * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": nbytes / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, "working_set_bytes": copies * len(seq),
            "what": f"1 B/base ASCII read + 0.375 B/base planes written, {copies} input buffers in rotation ({copies * len(seq) >> 20} MiB > Infinity Cache)"}


def chr1_record(bases: int) -> bytes:
    """SURVEY.md 8(d) cfg3 stand-in, largest record: generator seed 4, N blocks at both ends and one 3-Mbp
    centromere-like N block (record 0 of ribbit_amd.simulate.grch38_shaped_record, the definition the committed
    digests of tests/golden/grch38_shape_digests.json were made from)"""
    from ribbit_amd.simulate import grch38_shaped_record
    return grch38_shaped_record(0, bases, M_LO, M_HI)


def verify_chr1_digest(bases: int, bed_rows: int, bed_sha: str) -> dict:
    """-> {"verified": true / false / null, "verification": {...}}: the BED text of the chr1_full_path leg against the committed
    digest of the oracle pipeline's BED for the same record (tests/golden/grch38_shape_digests.json, record "chr1": generator
    seed 4 + N blocks, 248,956,422 bases, -m 2 -M 100).  null: no digest for this size (--chr1-bases other than the full one)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "grch38_shape_digests.json")
    try:
        want = json.load(open(path))["records"]["chr1"]
    except (OSError, KeyError, ValueError) as e:
        return {"verified": None, "verification": {"what": f"no digest available ({type(e).__name__})"}}
    if want["bases"] != bases or (want["m_lo"], want["m_hi"]) != (M_LO, M_HI):
        return {"verified": None, "verification": {"what": f"the committed digest is for {want['bases']} bases, -m {want['m_lo']} -M {want['m_hi']}; this run: {bases} bases"}}
    ok = bool(want["sha256"] == bed_sha and want["bed_rows"] == bed_rows)
    return {"verified": ok, "verification": {"bed_sha256_oracle": want["sha256"], "bed_rows_oracle": want["bed_rows"], "oracle_seconds_one_core": want["oracle_seconds"],
                                             "what": "SHA-256 and row count of the whole BED text == the CPU oracle pipeline's for the same 248,956,422-base record "
                                                     "(tests/golden/grch38_shape_digests.json, made by tests/golden/make_full_size_digests.py; the oracle is a "
                                                     "restatement of the reference: parity unpinned, DESIGN.md 2)"}}


def m500_full_path(ribbit_amd, bases: int, device: int):
    """BASELINE.json configs[4]'s motif range as stated (-m 2 -M 500: 499 composed planes, motifs beyond 128 bases) on a bounded record,
    through the whole path to the BED text, checked against the oracle pipeline's committed digest (tests/golden/
    grch38_shape_digests.json, entry "m500": 16 Mbp, 6.5 minutes of one core in the build container).  One pass, outside the headline."""
    import hashlib
    import numpy as np
    from ribbit_amd.simulate import m500_record
    seq = m500_record(bases)
    with ribbit_amd.Scanner(2, 500, device=device) as sc:
        t0 = time.perf_counter()
        sc.load_record(seq)
        sc.processShiftXORsAnchored(copy=False)
        dispatch = sc.dispatch_seeds(copy=False)
        t1 = time.perf_counter()
        bed = sc.refine_bed_view("m500")
        t2 = time.perf_counter()
        rows = int(np.count_nonzero(bed == ord("\n")))
        sha = hashlib.sha256(np.ascontiguousarray(bed)).hexdigest()
        n_dispatch = int(len(dispatch))
        del bed
    out = {"bases": bases, "min_motif": 2, "max_motif": 500, "seconds": t2 - t0, "value": bases / (t2 - t0) / 1e9, "unit": "Gbases/s",
           "scans_and_merges_s": t1 - t0, "refinement_and_bed_s": t2 - t1, "dispatched": n_dispatch, "bed_rows": rows, "bed_sha256": sha,
           "what": "one record at BASELINE.json configs[4]'s motif range, FASTA record -> BED text, first use of the handle (allocations included)"}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "grch38_shape_digests.json")
    try:
        want = json.load(open(path))["m500"]
    except (OSError, KeyError, ValueError) as e:
        out.update({"verified": None, "verification": {"what": f"no digest available ({type(e).__name__})"}})
        return out
    if want["bases"] != bases:
        out.update({"verified": None, "verification": {"what": f"the committed digest is for {want['bases']} bases; this run: {bases}"}})
        return out
    out.update({"verified": bool(want["sha256"] == sha and want["bed_rows"] == rows),
                "verification": {"bed_sha256_oracle": want["sha256"], "bed_rows_oracle": want["bed_rows"], "oracle_seconds_one_core": want["oracle_seconds"],
                                 "what": "SHA-256 and row count of the BED text == the CPU oracle pipeline's for the same record at -m 2 -M 500 "
                                         "(tests/golden/grch38_shape_digests.json: m500; parity unpinned, DESIGN.md 2)"}})
    return out


def chr1_full_path(ribbit_amd, bases: int, device: int, traffic: dict):
    """BASELINE.json configs[2] on its largest record: perfect + substitution + anchored scans, merges, dispatch,
    refinement, BED text.  Two passes over the scans and merges (the first includes every allocation), one refinement."""
    t_gen = time.perf_counter()
    seq = chr1_record(bases)
    t_gen = time.perf_counter() - t_gen
    out = {"bases": bases, "generator": "ribbit_amd.simulate seed 4 + N blocks (ends, 3 Mbp centre)", "generate_s": t_gen}
    with ribbit_amd.PinnedBuffer(bases) as buf, ribbit_amd.Scanner(M_LO, M_HI, device=device) as sc:
        import numpy as np
        buf.array[:] = np.frombuffer(seq, dtype=np.uint8)
        passes = []
        for rep in range(2):
            t0 = time.perf_counter()
            sc.load_record_pinned(buf.ptr, bases)
            # copy=False: the lists as the C ABI hands them out (pointers into the library's memory); copying half a
            # gigabyte of seeds into numpy arrays is the Python mirror's business, not the path's
            perfect_seeds = sc.processShiftXORsPerfect(copy=False)
            t1 = time.perf_counter()
            perfect, subst, anchored = sc.processShiftXORsAnchored(copy=False)
            dispatch = sc.dispatch_seeds(copy=False)
            t2 = time.perf_counter()
            n_lists = (int(len(perfect)), int(len(subst)), int(len(anchored)), int(len(dispatch)))
            on_device, left_to_host, host_meanwhile, ranges, again = ribbit_amd.last_device_merge()
            passes.append({"load_and_perfect_stage_s": t1 - t0, "substitution_and_anchored_stages_s": t2 - t1, "scans_and_merges_s": t2 - t0,
                           "merge_ms": {"substitution": sc.timing_ms(5), "anchored": sc.timing_ms(4)},
                           # the anchored stage's merge: ranges its first pass merged on the GPU (one lane each), on the host threads while
                           # that kernel ran, and after it (ranges the lanes gave up); all zero = the host threads alone
                           "anchored_merge_ranges": {"all": ranges, "on_the_gpu": on_device, "on_host_threads_meanwhile": host_meanwhile,
                                                     "left_by_the_gpu": left_to_host, "merged_again_by_the_walk": again}})
        kern = {"pack_kernel": sc.timing_ms(0), "scan_window_kernel<1>": sc.timing_ms(6), "scan_anchored_kernel": sc.timing_ms(7)}
        try:        # the anchored stage as two kernels: planes (anchors + composition), then the window scan of the planes
            kern["scan_anchored_kernel<planes>"], kern["scan_xa_window_kernel"] = sc.timing_ms(8), sc.timing_ms(9)
        except ribbit_amd.RibbitHipError:
            pass
        sc.scan_perfect_runs()
        kern["scan_perfect_kernel"] = sc.timing_ms(1)
        # refinement of the lists just made (the perfect re-scan above does not touch them)
        # (the text as the C ABI returns it, by pointer: what ribbit-hip writes to its output file; the Python mirror's copy and
        # decoding of 150 MB into a str is not the path's, like the seed lists above)
        t3 = time.perf_counter()
        bed = sc.refine_bed_view("chr1")
        t4 = time.perf_counter()
        bed_rows = int(np.count_nonzero(bed == ord("\n")))
        # full-size identity, outside the timed region: the SHA-256 of the BED text against the digest the ORACLE pipeline
        # (oracle/, CPU, tests/golden/make_full_size_digests.py: 18 minutes on one core of the build container) wrote for
        # exactly this record
        import hashlib
        bed_sha = hashlib.sha256(np.ascontiguousarray(bed)).hexdigest()
        del bed
    scans = passes[1]["scans_and_merges_s"]
    out.update({
        "passes": passes, "kernel_ms": kern, "scans_and_merges_s": scans, "refinement_and_bed_s": t4 - t3,
        "seconds": scans + (t4 - t3), "value": bases / (scans + (t4 - t3)) / 1e9, "unit": "Gbases/s",
        "scans_and_merges_gbases_per_s": bases / scans / 1e9,
        "seeds": {"perfect": n_lists[0], "substitution": n_lists[1], "anchored": n_lists[2]},
        "dispatched": n_lists[3], "bed_rows": bed_rows, "bed_sha256": bed_sha,
        "pass_1_scans_and_merges_s": passes[0]["scans_and_merges_s"], "pass_2_scans_and_merges_s": passes[1]["scans_and_merges_s"],
        "cold_seconds": passes[0]["scans_and_merges_s"] + (t4 - t3),
        "what": "one chromosome-1-sized record, -m 2 -M 100: FASTA record in page-locked memory -> BED text.  `seconds` = pass 2 of the scans and "
                "merges (buffers in place) + refinement; `cold_seconds` = pass 1 (every allocation, what a cold ribbit-hip pays) + refinement; "
                "the seed lists and the BED text are taken as the C ABI returns them, by pointer"})
    out.update(verify_chr1_digest(bases, bed_rows, bed_sha))
    roof = {}
    for name, key in (("scan_window_kernel<1>", "scan_window_kernel"), ("scan_anchored_kernel", "scan_anchored_kernel"), ("scan_perfect_kernel", None)):
        # "scan_anchored_kernel": the anchored stage's scan, i.e. its planes kernel and the window scan of the planes together
        ms = kern[name]
        achieved = bases * ALGO_BYTES_PER_BASE / (ms * 1e-3) / 1e9
        r = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
             "kernel_ms": ms, "algorithmic_bytes_per_launch": bases * ALGO_BYTES_PER_BASE, "kernel_gbases_per_s": bases / (ms * 1e-3) / 1e9, "traffic": None}
        if key == "scan_anchored_kernel" and traffic and traffic.get("anchored_stage_scan_hbm_bytes_per_base") is not None:
            key = "anchored_stage_scan"          # the stage's scan is two kernels (planes, window scan of the planes): their sum
            r["kernels"] = traffic.get("anchored_stage_scan_kernels")
        if key and traffic and traffic.get(key + "_hbm_bytes_per_base") is not None:
            # PMC passes were taken on a 100-Mbp record (profiles/): bytes per base carry over, the record is all that differs
            r["traffic"] = traffic[key + "_hbm_bytes_per_base"] * bases
            r["traffic_over_algorithmic"] = traffic[key + "_hbm_bytes_per_base"] / ALGO_BYTES_PER_BASE
            if traffic.get(key + "_hbm_write_bytes_per_base") is not None:
                r["hbm_write_bytes_per_base"] = traffic[key + "_hbm_write_bytes_per_base"]
            r["traffic_source"] = f"profiles/{traffic.get('tag', '?')} PMC passes (FETCH_SIZE x correction + WRITE_SIZE) per base x this record"
            if traffic.get(key + "_SQ_INSTS_VALU_per_base") is not None:
                r["valu_wave_instr_per_base"] = traffic[key + "_SQ_INSTS_VALU_per_base"]
        roof[name] = r
    out["roofline"] = roof
    return out


def full_path_sample(sc, seq: bytes, bases: int):
    """The whole per-sequence path of BASELINE.json configs[2] (perfect + substitution + anchored scans, the three
    seed merges, dispatch, refinement, BED text) on a bounded sample of the workload: reported next to the scan
    metric because it is what an end-to-end run waits for (host merges and refinement)."""
    sample = seq[:bases]
    t0 = time.perf_counter()
    sc.load_record(sample)
    perfect, subst, anchored = sc.processShiftXORsAnchored()
    dispatch = sc.dispatch_seeds()
    t1 = time.perf_counter()
    bed = sc.refine_bed("bench")
    t2 = time.perf_counter()
    # the same path through the CPU oracle (single thread) on a smaller sample, for scale
    from oracle_lib import Oracle
    small = seq[:min(len(sample), CPU_FULL_PATH_BASES)]
    c0 = time.perf_counter()
    with Oracle(small, M_LO, M_HI) as o:
        o.run_all()
        o.refine_bed("bench")
    cpu_dt = time.perf_counter() - c0
    return {"bases": len(sample), "value": len(sample) / (t2 - t0) / 1e9, "unit": "Gbases/s", "seconds": t2 - t0,
            "cpu_port": {"value": len(small) / cpu_dt / 1e9, "unit": "Gbases/s", "cores": 1, "bases": len(small), "seconds": cpu_dt},
            "scans_and_merges_s": t1 - t0, "refinement_and_bed_s": t2 - t1, "seeds": int(len(perfect) + len(subst) + len(anchored)),
            "dispatched": int(len(dispatch)), "bed_rows": bed.count("\n"),
            "what": "FASTA record -> BED text, -m 2 -M 100: three scans + seed merges + dispatch + refinement (host-bound)"}


def sharded_full_path(ribbit_amd, dist, torch, sc, seq: bytes, bases: int, rank: int, world: int, xdev, dev):
    """N > 1: BASELINE.json configs[3]'s shape on a bounded record -- ONE record of world x `bases` chunk-sharded over the
    ranks, the whole P+S+A path: every rank runs the device side of the three stages on its chunk (scan kernels, pairing,
    window state machines, length filters, call order) and keeps the calls whose scan position it owns; 16 bytes per kept
    call and the rank's plane words travel to rank 0 (gather-v over the collective backend: RCCL on GPUs), which runs the
    three merges and the dispatch merge once (ribbit_amd.sharded).  Rank 0 then scans the whole record on its own GPU and
    checks that the sharded lists are identical.  One pass, outside the headline's timed region."""
    import numpy as np
    from ribbit_amd import sharded
    from ribbit_amd.distributed import allgather_array
    chunks = allgather_array(np.frombuffer(seq[:bases], dtype=np.uint8).copy(), xdev)     # every rank can fetch any halo
    record = b"".join(c.tobytes() for c in chunks)
    L = len(record)
    plan = sharded.plan_chunks(L, world, M_HI)[rank]

    def sync():
        dist.barrier()
        torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    part = sharded.scan_part(sc, record, plan)
    torch.cuda.synchronize()
    t_scan = time.perf_counter() - t0
    # the exchange, over both transports when the ranks share a node (as the headline leg does): the collective backend
    # (RCCL on GPUs: host -> device -> gather -> host, since what a chunk keeps is host data by now), and the node-shared segment
    # (every rank copies into its own cell: N memcpys side by side).  The leg's total uses the faster one; both are in the line.
    from ribbit_amd.distributed import same_node
    backend = "rccl" if xdev is not None else "gloo"
    sync()
    t1 = time.perf_counter()
    gathered = sharded.gather_parts(part, xdev)
    sync()
    by_transport = {backend: time.perf_counter() - t1}
    shm_identical = None
    if same_node():
        sync()
        t1 = time.perf_counter()
        ok, shared = sharded.gather_parts_shm(part)
        sync()
        if ok:
            by_transport["shm"] = time.perf_counter() - t1
            if rank == 0:
                shm_identical = all((a[k] is None) == (b[k] is None) and (a[k] is None or np.ndim(a[k]) == 0 and a[k] == b[k] or
                                                                            np.ndim(a[k]) > 0 and np.asarray(a[k]).tobytes() == np.asarray(b[k]).tobytes())
                                    for a, b in zip(gathered, shared) for k in a)
    headline_transport = min(by_transport, key=by_transport.get)
    t_exchange = by_transport[headline_transport]
    t = torch.tensor([t_scan], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    sent = sharded.part_bytes(part)
    all_sent = [None] * world
    dist.all_gather_object(all_sent, dict(sent, halo_grown=part["halo_grown"], left_halo=part["left_halo"]))
    lists, t_merge = None, 0.0
    if rank == 0:
        t2 = time.perf_counter()
        lists = sharded.merge_parts(M_LO, M_HI, L, gathered)
        t_merge = time.perf_counter() - t2
    # Refinement over the ranks (round 4; ribbit_hip_adopt_dispatch): the dispatched seeds are independent, so rank r takes the r-th
    # slice of the dispatch list -- 16 bytes a seed from rank 0 -- with the record loaded on its own GPU (the composed planes made
    # there by the planes kernel alone), refines it on its GPU and its share of the host threads, and the BED text of the slices goes
    # back to rank 0 in order.  (An alignment with an empty query at the head of a slice would need the record refined in one
    # piece, as ribbit-hip does: reported here, and the texts' identity with the one-GPU run is checked either way.)
    sync()
    t3 = time.perf_counter()
    box = [lists["dispatch"] if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    dispatch_all = box[0]
    nd = len(dispatch_all)
    lo, hi = nd * rank // world, nd * (rank + 1) // world
    sc.load_record(record)
    sc.adopt_dispatch(dispatch_all[lo:hi])
    my_bed = sc.refine_bed("rec")
    met_empty = sc.refine_met_empty_query()
    torch.cuda.synchronize()
    t_refine_own = time.perf_counter() - t3
    beds = [None] * world
    dist.all_gather_object(beds, (my_bed, bool(met_empty), t_refine_own))
    sync()
    t_refine = time.perf_counter() - t3
    if rank != 0:
        return None
    sc.load_record(record)
    perfect, subst, anchored = sc.processShiftXORsAnchored()
    dispatch = sc.dispatch_seeds()
    same = all(np.array_equal(lists[k].view("<i4"), w.view("<i4")) for k, w in
               (("perfect", perfect), ("subst", subst), ("anchored", anchored), ("dispatch", dispatch)))
    t4 = time.perf_counter()
    bed_one = sc.refine_bed("rec")
    t_refine_one_gpu = time.perf_counter() - t4
    bed_sharded = "".join(b[0] for b in beds)
    any_empty = any(b[1] for b in beds[1:])
    total = float(t.item()) + t_exchange + t_merge + t_refine
    return {"bases": L, "bases_per_gpu": bases, "seconds": total, "value": L / total / 1e9, "unit": "Gbases/s",
            "scan_seconds_max_over_ranks": float(t.item()), "exchange_seconds": t_exchange, "merge_seconds_rank0": t_merge,
            "refinement": {"seconds": t_refine, "seconds_per_rank": [b[2] for b in beds], "seconds_on_one_gpu": t_refine_one_gpu,
                           "bed_rows": bed_sharded.count("\n"), "identical_to_single_gpu_bed": bool(bed_sharded == bed_one),
                           "empty_query_at_a_slice_head": bool(any_empty),
                           "what": "the dispatch list from rank 0 (16 B a seed), a slice per rank refined on that rank's GPU and host threads "
                                   "(ribbit_hip_adopt_dispatch), the slices' BED text gathered on rank 0; `seconds` from the broadcast to the last text"},
            "exchange": {"headline": headline_transport, "headline_rule": "the faster of the transports timed in this run",
                         "seconds": by_transport, "shm_identical_to_collective": shm_identical},
            "identical_to_single_gpu_scan": bool(same),
            "seeds": {"perfect": int(len(lists["perfect"])), "substitution": int(len(lists["subst"])), "anchored": int(len(lists["anchored"]))},
            "dispatched": int(len(lists["dispatch"])), "sent_per_rank": all_sent,
            "what": "one record chunk-sharded over the ranks, perfect + substitution + anchored stages: per-rank device stages (scan, pairing, "
                    "window state machines, filters), gather-v of 16-byte run records and kept calls + the ranks' plane words to rank 0, "
                    "merges and dispatch order on rank 0, refinement sharded by dispatched seed over the ranks"}


def main():
    global M_HI
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bases", type=int, default=WORKLOAD_BASES, help="bases per GPU (default: BASELINE config 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline and the full-path sample")
    ap.add_argument("--full-path-bases", type=int, default=10_000_000,
                    help="size of the whole-path sample reported next to the scan metric (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "shm", "rccl"],
                    help="N > 1: how the chunks' run records reach rank 0's host merge (auto: shm on one node, else rccl)")
    ap.add_argument("--depth", type=int, default=3,
                    help="batches in flight per GPU (handles/streams alternating); 1 = strictly one after the other")
    ap.add_argument("--streams", default="shared", choices=["shared", "own"],
                    help="shared: one compute stream for all handles of a rank (clean per-kernel timings); own: one per handle")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--verify", action="store_true",
                    help="N > 1: rank 0 also scans the whole record on its own GPU and checks the sharded runs against it")
    ap.add_argument("--calibrate", action="store_true",
                    help="after the timed region, launch the known-byte-count stream-read kernel (for PMC passes)")
    ap.add_argument("--chr1-bases", type=int, default=CHR1_BASES,
                    help="size of the BASELINE configs[2] record of the chr1_full_path leg (0 = skip the leg)")
    ap.add_argument("--m500-bases", type=int, default=16_000_000,
                    help="size of the -M 500 record of the m500_full_path leg (BASELINE configs[4]'s motif range; 0 = skip the leg)")
    ap.add_argument("--max-motif", type=int, default=M_HI,
                    help="-M of the run (default: BASELINE configs[1]'s 100).  Anything else is a profiling run -- tools/profile_bench.sh "
                         "with 500 for configs[4]'s motif range -- and its line says so in config.workload; the headline is quoted at the default")
    ap.add_argument("--stage-kernels", action="store_true",
                    help="after the timed region, run the substitution and anchored scan kernels once on the workload record "
                         "(so that a profiler pass over this command sees all three scan kernels at the same size)")
    args = ap.parse_args()
    if args.max_motif != M_HI:
        M_HI = args.max_motif
        args.chr1_bases = 0          # the chromosome leg and its digest are for -M 100

    # --gpus N without a launcher: start one process per GPU ourselves -- as a CHILD, before this process has touched a
    # GPU (an exec after HIP is initialised takes the machine down on this pool) -- and leave with its exit code.
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} processes")

    import numpy as np
    import torch

    import ribbit_amd
    from ribbit_amd.distributed import allgather_array, gather_array, open_node_gather, same_node
    from ribbit_amd.simulate import simulate_sequence

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: ribbit_amd has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend=args.backend)

    # synthetic record: rank r owns chunk r (generator seed 2 + r) of one record of world x bases
    seq, _ = simulate_sequence(args.bases, 2 + rank, M_LO, M_HI)
    dev = torch.device("cuda", local_rank)
    xdev = dev if args.backend == "nccl" else None          # where the exchange buffers live
    halo_l = halo_r = b""
    if world > 1:
        # halos from the neighbours (a few hundred bases; exchanged once, outside the timed region)
        reach = 4 * (M_HI + 2) + 64
        edges = allgather_array(np.frombuffer(seq[:reach] + seq[-reach:], dtype=np.uint8).copy(), xdev)
        if rank > 0:
            halo_l = edges[rank - 1][reach:].tobytes()[-(2 * (M_HI + 2) + 64) // 32 * 32:]
        if rank < world - 1:
            halo_r = edges[rank + 1][:reach].tobytes()
    loaded = halo_l + seq + halo_r
    own_lo = len(halo_l)
    own_hi = own_lo + args.bases + (1 if rank == world - 1 else 0)     # the last chunk owns the end-of-record position
    d_ascii = torch.frombuffer(bytearray(loaded), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()

    # Two handles (each with its own HIP stream) alternate: while one batch's run records cross PCIe, the next
    # batch is already being packed and scanned -- the double-buffered streaming of a multi-record FASTA.
    depth = max(1, args.depth)
    scs = [ribbit_amd.Scanner(M_LO, M_HI, device=local_rank) for _ in range(depth)]
    sc = scs[0]
    # All handles launch on ONE compute stream (kernels of consecutive batches run back to back, so a kernel's
    # HIP-event duration is its own, not inflated by a neighbour's kernels); only the result copies, which each
    # handle issues on its own copy stream, overlap the next batch's kernels.
    compute = torch.cuda.Stream(device=dev) if args.streams == "shared" else None
    if compute is not None:
        for h in scs:
            h.set_stream(compute.cuda_stream)
    # Kernel durations come from the HIP events of handle 0 only, i.e. of every depth-th launch of the timed region,
    # evenly interleaved with the others: each event record is a barrier packet between kernels (together ~10 % of a
    # step), and one handle's launches are as good a sample of the kernel as all of them.
    timed_handle = [scs[0]]
    pos_offset = rank * args.bases - own_lo
    lo_hi = (own_lo, own_hi) if world > 1 else (0, (1 << 63) - 1)

    # N > 1: every rank scans and pairs its chunk on its GPU; the run records (16 B each: the candidate seed intervals)
    # are then gathered for the host-side merge on rank 0.  Transports:
    #   rccl  the records stay in HBM and travel GPU-to-GPU over xGMI (grouped send / recv through RCCL) to rank 0's
    #         GPU, then down rank 0's PCIe link once (ribbit_amd.distributed.DeviceGather) -- what north_star names;
    #   shm   (one node) every GPU copies its records down its OWN PCIe link into a page-locked segment all ranks map:
    #         N links in parallel, no second hop.
    # Both are timed in every N > 1 run on one node and both are in the line (`exchange`); `value` is the faster one's.  By
    # bandwidth arithmetic that should be shm from a few GPUs on (N x 10.6 MB per step through rank 0's one link against N
    # links side by side) -- arithmetic, not a measurement: the development box has one GPU.
    transports = ["local"]
    if world > 1:
        first = "rccl" if args.exchange == "auto" else args.exchange
        transports = [first] + (["shm"] if args.exchange == "auto" and same_node() else [])
    device_gather = world > 1 and args.backend == "nccl"      # gloo rehearsals gather host arrays instead
    T = {"name": transports[0], "ng": None, "dg": None}

    def setup(name):
        T["name"], T["ng"] = name, None
        if name == "rccl" and device_gather and T["dg"] is None:
            from ribbit_amd.distributed import DeviceGather
            T["dg"] = DeviceGather(dev)
        if name == "shm":
            sc.load_record_device(d_ascii.data_ptr(), d_ascii.numel())
            probe, _ = sc.scan_perfect_chunk(own_lo, own_hi, pos_offset)
            cap = max(int(a[0]) for a in allgather_array(np.array([len(probe)], dtype=np.int64), xdev))
            ng = open_node_gather(ribbit_amd.RUN_DT, 2 * cap + 4096, 2 * (M_HI - M_LO + 1), nslots=depth + 1)
            if ng is None:
                if rank == 0:
                    print("node-shared segment unavailable: gathering over RCCL instead", file=sys.stderr)
                T["name"] = "rccl"
                return setup("rccl")
            T["ng"] = ng
            for addr, nbytes in ng.my_cells():
                try:
                    sc.host_register(addr, nbytes)
                except ribbit_amd.RibbitHipError as e:       # still correct, the copies are just staged by the runtime
                    print(f"rank {rank}: shared segment not page-locked ({e})", file=sys.stderr)
                    break

    def teardown():
        ng = T["ng"]
        if ng is not None:
            dist.barrier()
            for addr, _ in ng.my_cells():
                try:
                    sc.host_unregister(addr)
                except ribbit_amd.RibbitHipError:
                    pass
            ng.close()
            T["ng"] = None

    def issue(k):
        """enqueue batch k: pack + scan + pairing kernels on handle k % depth, no waiting"""
        h = scs[k % depth]
        h.load_record_device(d_ascii.data_ptr(), d_ascii.numel())
        h.scan_perfect_begin(lo_hi[0], lo_hi[1], pos_offset if world > 1 else 0)

    def collect(k):
        """batch k's kernels are done: start moving its run records (no waiting); -> what finish() needs"""
        h = scs[k % depth]
        ng = T["ng"]
        if world > 1 and T["name"] == "rccl" and device_gather:
            return h.scan_perfect_end_device()            # nothing is copied: the records stay in HBM
        if world == 1 or ng is None:
            return h.scan_perfect_end(wait=False)
        if rank == 0 and k > 1:
            ng.release(k - 1)                             # the previous batch's views are dead from here on
        ng.wait_free(k)
        rec, hv = ng.mine(k)
        n, nh = h.scan_perfect_end(out=rec, halves_out=hv, wait=False)
        return rec, n, nh

    def finish(k, pending):
        """batch k's run records on the host (rank 0: of every rank)"""
        h = scs[k % depth]
        ng = T["ng"]
        if world > 1 and T["name"] == "rccl" and device_gather:
            ptr, n, hptr, nh = pending
            all_runs, all_halves = T["dg"].gather(ptr, n, hptr, nh, ribbit_amd.RUN_DT)
            if rank == 0:
                return all_runs + [ribbit_amd.join_run_halves(all_halves)]
            return [np.zeros(n, ribbit_amd.RUN_DT)]       # (this rank's records went to rank 0; only their number is of use here)
        h.scan_perfect_wait()
        if world == 1:
            return pending[0]                             # view of the C ABI's own pinned result buffer
        if ng is not None:
            rec, n, nh = pending
            ng.publish(k, n, nh)
            if rank != 0:
                return [rec[:n]]
            parts, halves = ng.collect(k)
            # the record's runs: every chunk's records in place (term < 0 = place holder) + the runs cut by chunk edges
            return parts + [ribbit_amd.join_run_halves(halves)]
        runs, halves = pending
        all_runs = gather_array(runs, xdev)
        all_halves = gather_array(halves, xdev)
        if rank == 0:
            return all_runs + [ribbit_amd.join_run_halves(all_halves)]
        return [runs]

    batch = [0]          # batches are numbered 1, 2, ... across warm-up and timed region

    def run_steps(n, on_step=None):
        first, last = batch[0] + 1, batch[0] + n
        issued, out = first - 1, None
        for k in range(first, min(first + depth - 2, last) + 1):     # prologue: depth - 1 batches in flight
            issued += 1
            issue(issued)
        for k in range(first, last + 1):
            pending = collect(k) if issued >= k else None
            if issued < last:                             # next batch's kernels are enqueued while k's results travel
                issued += 1
                issue(issued)
            if pending is None:
                pending = collect(k)
            out = finish(k, pending)
            if on_step:
                on_step(scs[k % depth], out)
        batch[0] = last
        return out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(name):
        """warm-up, then exactly args.steps timed steps of transport `name`: -> (seconds, max over ranks), last batch's runs, ..."""
        setup(name)
        batch[0] = 0
        run_steps(args.warmup)
        k_ms, p_ms, g_ms, cnt = [], [], [], [0, 0]

        def on_step(h, out):
            if h is timed_handle[0]:
                k_ms.append(h.timing_ms(1))
                p_ms.append(h.timing_ms(0))
                if not (world > 1 and T["name"] == "rccl" and device_gather):
                    g_ms.append(h.timing_ms(2))
            cnt[0] = len(out) if world == 1 else sum(len(r) for r in out)
            cnt[1] = h.last_event_count()

        timed_handle[0] = scs[(batch[0] + 1) % depth]    # the handle of the first timed batch, then of every depth-th
        for h in scs:
            h.set_timing(h is timed_handle[0])
        fence()
        t0 = time.perf_counter()
        last_runs = run_steps(args.steps, on_step)
        fence()
        seconds = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([seconds], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            seconds = float(t.item())
        used = T["name"]
        return seconds, last_runs, k_ms, p_ms, g_ms, cnt, used

    # The RCCL path cannot be rehearsed on the one-GPU development box (RCCL refuses two ranks on one device), so it is
    # probed first: one step, and a vote over a gloo side channel.  If any rank failed, every rank falls back to the
    # node-shared segment and the line says so.
    rccl_error = None
    if world > 1 and transports[0] == "rccl" and device_gather:
        ctl = dist.new_group(backend="gloo")
        ok = 1
        try:
            setup("rccl")
            batch[0] = 0
            run_steps(1)
            torch.cuda.synchronize()
        except Exception as e:      # noqa: BLE001 -- whatever the collective layer raises
            ok, rccl_error = 0, f"{type(e).__name__}: {e}"[:300]
        votes = [None] * world
        dist.all_gather_object(votes, (ok, rccl_error), group=ctl)
        if not all(v[0] for v in votes):
            rccl_error = next(v[1] for v in votes if not v[0])
            if rank == 0:
                print(f"RCCL gather of the device-resident records failed ({rccl_error}); using the node-shared segment", file=sys.stderr)
            transports = ["shm"]
            device_gather = False
    dt, runs, kernel_ms, pack_ms, gpu_ms, counts, used_transport = timed_region(transports[0])
    nruns, nevents = counts
    headline_ng = T["ng"] is not None
    other = {}

    if args.calibrate:
        sc.debug_stream_read(256 << 20)

    if args.verify and world > 1:
        whole = allgather_array(np.frombuffer(seq, dtype=np.uint8).copy(), xdev)
        if rank == 0:
            got = np.concatenate(runs)
            got = np.sort(got[got["term"] >= 0], order=["mlen", "start"])
            sc.load_record(b"".join(w.tobytes() for w in whole))
            want = sc.scan_perfect_runs()
            assert np.array_equal(got.view("<i4"), want.view("<i4")), "chunk-sharded runs differ from the single-GPU scan"
            print(f"verify: {len(want)} runs identical to the single-GPU scan of the whole record", file=sys.stderr)

    first_transport, first_dt = used_transport, dt
    for name in transports[1:]:
        teardown()
        sec, _r, _k, _p, _g, _c, used = timed_region(name)
        other[used] = {"ms_per_step": sec / args.steps * 1e3, "value": args.bases * world * args.steps / sec / 1e9, "unit": "Gbases/s"}
        # `value` is the job's throughput: with both transports timed on the same ranks in the same run, it is that of the
        # faster one (rank 0 decides, every rank timed the same max-over-ranks seconds); both stay in `exchange`
        if sec < dt:
            other[first_transport] = {"ms_per_step": first_dt / args.steps * 1e3, "value": args.bases * world * args.steps / first_dt / 1e9, "unit": "Gbases/s"}
            del other[used]
            dt, used_transport = sec, used
    headline_ng = used_transport == "shm"
    headline_dg = used_transport == "rccl" and device_gather

    sharded_leg = None
    if world > 1 and args.full_path_bases > 0:
        teardown()
        sharded_leg = sharded_full_path(ribbit_amd, dist, torch, sc, seq, min(args.full_path_bases, args.bases), rank, world, xdev, dev)

    if rank == 0:
        traffic = n_valu = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.bases == WORKLOAD_BASES:
            # HBM bytes per scan_perfect_kernel launch from the committed rocprofv3 PMC passes
            # (FETCH_SIZE / WRITE_SIZE collected separately, corrected as profiles/README.md explains)
            prof = json.load(open(tpath))
            traffic = prof.get("scan_perfect_kernel_hbm_bytes_per_launch")
            n_valu = prof.get("scan_perfect_kernel_SQ_INSTS_VALU")
        total_bases = args.bases * world * args.steps
        kavg = float(np.mean(kernel_ms))
        achieved = args.bases * ALGO_BYTES_PER_BASE / (kavg * 1e-3) / 1e9
        out = {
            "metric": "Gbases/s scanned (m=2..100)", "value": total_bases / dt / 1e9, "unit": "Gbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 bit planes",
            "data": "synthetic (ribbit_amd.simulate, seeded restatement of data_simulation/simulate_data.py)",
            "config": {"workload": ("" if M_HI == 100 else "PROFILING RUN AT A NON-DEFAULT MOTIF RANGE: ") + f"{args.bases} bp synthetic record per GPU, -m {M_LO} -M {M_HI}, "
                                   "pack + perfect shift-XOR scan (BASELINE.json configs[1])",
                       "batches_in_flight": depth, "compute_streams": 1 if compute is not None else depth,
                       "kernel_timing": f"HIP events on every {depth}th launch of the timed region" if depth > 1 else "HIP events on every launch",
                       "bases_per_gpu": args.bases, "min_motif": M_LO, "max_motif": M_HI,
                       "parallelism": (f"one record chunk-sharded x{world} (halos), runs paired on each GPU, gathered for rank 0's host merge "
                                       + ("through page-locked node-shared memory (one PCIe link per GPU)" if headline_ng else
                                          ("by gather-v of the device-resident records over RCCL / xGMI (grouped send/recv)" if headline_dg else "by gather-v over the collective backend")))
                                      if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "scan_perfect_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": args.bases * ALGO_BYTES_PER_BASE,
                         "kernel_ms": kavg, "pack_kernel_ms": float(np.mean(pack_ms)),
                         "kernel_gbases_per_s": args.bases / (kavg * 1e-3) / 1e9,
                         "note": "integer-VALU bound by design (SURVEY.md 8d); HBM fraction reported as required"},
            "runs_per_step": nruns, "device_events_per_step": nevents,
            "gpu_side_ms_per_step": float(np.mean(gpu_ms)) if gpu_ms else None,   # scan + pairing kernels + D2H of the runs (HIP events)
        }
        if world > 1:
            out["exchange"] = {"headline": used_transport, "headline_rule": "the faster of the transports timed in this run",
                               used_transport: {"ms_per_step": dt / args.steps * 1e3, "value": total_bases / dt / 1e9, "unit": "Gbases/s"}}
            out["exchange"].update(other)
            if rccl_error:
                out["exchange"]["rccl_error"] = rccl_error
            if sharded_leg is not None:
                out["full_path_sharded"] = sharded_leg
        if n_valu:
            # the honest limiter: wave-instructions per launch from the committed SQ_INSTS_VALU pass, the share of
            # v_alignbit among them from profiles/isa_mix.json, against the issue rates measured on this part by
            # tools/probes/valu_peak.hip at the kernel's occupancy
            share, bracket, source = alignbit_share()
            roof_ms = (n_valu * share / ALIGNBIT_RATE + n_valu * (1 - share) / PLAIN_VALU_RATE) * 1e3
            out["roofline"]["valu"] = {"wave_instr_per_launch": n_valu, "alignbit_share": share, "alignbit_share_bracket": bracket,
                                       "alignbit_share_source": source,
                                       "issue_rate_wave_instr_per_s": {"v_alignbit_b32": ALIGNBIT_RATE, "other": PLAIN_VALU_RATE},
                                       "issue_bound_ms": roof_ms, "frac": roof_ms / kavg}
        if args.stage_kernels:
            sc.load_record_device(d_ascii.data_ptr(), d_ascii.numel())
            sc.processShiftXORsAnchored()
            out["stage_kernels_ms"] = {"scan_window_kernel<1>": sc.timing_ms(6), "scan_anchored_kernel": sc.timing_ms(7)}
            try:
                out["stage_kernels_ms"].update({"scan_anchored_kernel<planes>": sc.timing_ms(8), "scan_xa_window_kernel": sc.timing_ms(9)})
            except ribbit_amd.RibbitHipError:
                pass
        if world == 1 and not args.no_cpu_baseline:
            prof = json.load(open(tpath)) if os.path.exists(tpath) else {}
            cpu, oracle_calls, oracle_seeds = cpu_baseline(seq)
            ok, what = verify_against_oracle(sc, seq, oracle_calls, oracle_seeds)
            out["verified"] = ok
            out["verification"] = what
            out["pcie_inclusive"] = pcie_inclusive(ribbit_amd, seq, max(4, min(args.steps, 12)), depth, local_rank)
            out["pack_hbm"] = pack_hbm(ribbit_amd, torch, seq, dev, local_rank)
            if args.full_path_bases > 0:
                out["full_path_sample"] = full_path_sample(sc, seq, min(args.full_path_bases, args.bases))
            if args.chr1_bases > 0:
                for h in scs[1:]:
                    h.close()                     # their buffers are not needed any more; the chr1 record wants the room
                out["chr1_full_path"] = chr1_full_path(ribbit_amd, args.chr1_bases, local_rank, prof)
            if args.m500_bases > 0 and M_HI == 100:
                for h in scs[1:]:
                    h.close()
                out["m500_full_path"] = m500_full_path(ribbit_amd, args.m500_bases, local_rank)
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)

    runs = None
    teardown()
    for h in scs:
        h.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
