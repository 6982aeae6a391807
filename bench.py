#!/usr/bin/env python3
"""bench.py -- Gbases/s of the shift-XOR scan (BASELINE.json metric) on N MI355X GPUs.

One "step" = one pass of the hot path over one record that is already resident in HBM as ASCII:
pack kernel (fasta_utils.cpp:78-115) + perfect shift-XOR scan kernel over m=2..100
(fasta_utils.cpp:117-122 + parse_perfect_shiftxor.cpp:173-223) + event read-back + host pairing
into runs.  Workload = BASELINE.json configs[1]: 100 Mbp synthetic FASTA, -m 2 -M 100, perfect scan.

N > 1: one process per GPU (torch.distributed, backend nccl == RCCL).  The path shards by record
(option 1): every rank scans its own 100-Mbp record (weak scaling, no data-path
collective) and the sparse run records are exchanged with an all-gather-v over RCCL before the
(host) merge, as BASELINE.json's north_star prescribes.

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
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s
CPU_SAMPLE_BASES = 20_000_000


def cpu_baseline(seq: bytes):
    """Oracle ("port") timed on this box's host cores, single thread, on a bounded sample."""
    from oracle_lib import Oracle
    sample = seq[:CPU_SAMPLE_BASES]
    t0 = time.perf_counter()
    with Oracle(sample, M_LO, M_HI) as o:
        o.run_perfect()
    dt = time.perf_counter() - t0
    return {"value": len(sample) / dt / 1e9, "unit": "Gbases/s", "cores": 1, "kind": "port",
            "sample": f"first {len(sample)} bases of the workload, encode + sweep + perfect scan + addSeed, m={M_LO}..{M_HI}, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bases", type=int, default=WORKLOAD_BASES, help="bases per GPU (default: BASELINE config 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--calibrate", action="store_true",
                    help="after the timed region, launch the known-byte-count stream-read kernel (for PMC passes)")
    args = ap.parse_args()

    import numpy as np
    import torch

    import ribbit_amd
    from ribbit_amd.distributed import allgather_records
    from ribbit_amd.simulate import simulate_sequence

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: ribbit_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl")

    # synthetic records: every rank owns one record, generator seed 2 + rank
    seq, _ = simulate_sequence(args.bases, 2 + rank, M_LO, M_HI)
    dev = torch.device("cuda", local_rank)
    d_ascii = torch.frombuffer(bytearray(seq), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()

    sc = ribbit_amd.Scanner(M_LO, M_HI, device=local_rank)

    def step():
        sc.load_record_device(d_ascii.data_ptr(), d_ascii.numel())
        runs = sc.scan_perfect_runs()
        if world > 1:
            # all-gather-v of the sparse run records over RCCL/xGMI (count exchange + padded gather)
            allgather_records(runs, dev)
        return runs

    for _ in range(args.warmup):
        step()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    kernel_ms, pack_ms, nruns, nevents = [], [], 0, 0
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runs = step()
        kernel_ms.append(sc.timing_ms(1))
        pack_ms.append(sc.timing_ms(0))
        nruns, nevents = len(runs), sc.last_event_count()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if args.calibrate:
        sc.debug_stream_read(256 << 20)

    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.bases == WORKLOAD_BASES:
            # HBM bytes per scan_perfect_kernel launch from the committed rocprofv3 PMC passes
            # (FETCH_SIZE / WRITE_SIZE collected separately, corrected as profiles/README.md explains)
            traffic = json.load(open(tpath)).get("scan_perfect_kernel_hbm_bytes_per_launch")
        total_bases = args.bases * world * args.steps
        kavg = float(np.mean(kernel_ms))
        achieved = args.bases * ALGO_BYTES_PER_BASE / (kavg * 1e-3) / 1e9
        out = {
            "metric": "Gbases/s scanned (m=2..100)", "value": total_bases / dt / 1e9, "unit": "Gbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 bit planes",
            "data": "synthetic (ribbit_amd.simulate, seeded restatement of data_simulation/simulate_data.py)",
            "config": {"workload": f"{args.bases} bp synthetic record per GPU, -m {M_LO} -M {M_HI}, "
                                   "pack + perfect shift-XOR scan (BASELINE.json configs[1])",
                       "bases_per_gpu": args.bases, "min_motif": M_LO, "max_motif": M_HI,
                       "parallelism": f"record-sharded x{world} + all-gather-v of runs" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "scan_perfect_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": args.bases * ALGO_BYTES_PER_BASE,
                         "kernel_ms": kavg, "pack_kernel_ms": float(np.mean(pack_ms)),
                         "kernel_gbases_per_s": args.bases / (kavg * 1e-3) / 1e9,
                         "note": "integer-VALU bound by design (SURVEY.md 8d); HBM fraction reported as required"},
            "runs_per_step": nruns, "device_events_per_step": nevents,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(seq)
        print(json.dumps(out), flush=True)

    sc.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
