#!/usr/bin/env python3
"""BASELINE.json configs[4] as stated, on one GPU: `-m 2 -M 500` through the command-line front end.

    python tools/m500_probe.py record [--bases 248956422]        one chromosome-sized record (generator motifs 2..500, N blocks)
    python tools/m500_probe.py reads  [--total 1000000000]       a stream of 10-100 kb records (simulated long reads), seed 5
    options: --M 500  --jobs N  --keep  --dir /tmp/m500

Generation (process pool) and FASTA writing are outside the timed region.  Prints one JSON line: wall time, Mbases/s, BED
rows, the front end's per-stage sums (RIBBIT_PROFILE), the peak of the GPU's used memory (sysfs mem_info_vram_used, polled
every 50 ms while ribbit-hip runs) and ribbit-hip's peak resident host memory (getrusage of the child)."""
import argparse
import glob
import json
import os
import resource
import subprocess
import sys
import threading
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fasta_body(b: np.ndarray) -> bytes:
    n = len(b)
    full = n // 80 * 80
    body = np.empty((full // 80, 81), dtype=np.uint8)
    body[:, :80] = b[:full].reshape(-1, 80)
    body[:, 80] = ord("\n")
    return body.tobytes() + (b[full:].tobytes() + b"\n" if full < n else b"")


def make_record(args):
    from ribbit_amd.simulate import grch38_shaped_record
    bases, M, path = args
    b = np.frombuffer(grch38_shaped_record(0, bases, 2, M), dtype=np.uint8)
    with open(path, "wb") as fh:
        fh.write(b">chr1_M%d\n" % M)
        fh.write(fasta_body(b))
    return path


def make_reads(args):
    """one piece of the stream: `bases` bases of generator output (seed 5 + piece) cut into reads of U[10 kb, 100 kb]"""
    from ribbit_amd.simulate import simulate_sequence
    piece, bases, M, path = args
    seq, _ = simulate_sequence(bases, 5 + 7919 * piece, 2, M)
    b = np.frombuffer(seq, dtype=np.uint8)
    rs = np.random.RandomState(1000 + piece)
    at, k = 0, 0
    with open(path, "wb") as fh:
        while at < bases:
            n = int(min(bases - at, rs.randint(10_000, 100_001)))
            fh.write(b">read_%d_%d\n" % (piece, k))
            fh.write(fasta_body(b[at:at + n]))
            at += n; k += 1
    return path, k


class VramPeak:
    """per card of the host: used memory before the run and its peak during it.  The other cards belong to other jobs, so the figures
    reported are those of the card ribbit-hip says it ran on (its "[device] slot 0 is GPU n at PCI ..." line under RIBBIT_PROFILE)."""
    def __init__(self):
        self.files = glob.glob("/sys/class/drm/card*/device/mem_info_vram_used")
        self.stop = False
        self.before = {f: self.read(f) for f in self.files}
        self.top = dict(self.before)
        self.t = threading.Thread(target=self.loop, daemon=True)

    @staticmethod
    def read(f):
        try:
            return int(open(f).read())
        except (OSError, ValueError):
            return 0

    def loop(self):
        while not self.stop:
            for f in self.files:
                self.top[f] = max(self.top[f], self.read(f))
            time.sleep(0.05)

    def card_of(self, pci: str):
        for f in self.files:
            try:
                if os.path.basename(os.path.realpath(os.path.dirname(f))).lower() == pci.lower():
                    return f
            except OSError:
                pass
        return None

    def report(self, stderr_text: str) -> dict:
        import re
        m = re.search(r"\[device\] slot 0 is GPU \d+ at PCI (\S+)", stderr_text)
        f = self.card_of(m.group(1)) if m else None
        if f is None:
            return {"peak_vram_gb": None, "vram_note": "the run's card was not identified among the host's cards"}
        return {"peak_vram_gb": round(self.top[f] / 1e9, 2), "vram_before_gb": round(self.before[f] / 1e9, 2),
                "vram_grown_gb": round((self.top[f] - self.before[f]) / 1e9, 2), "card": m.group(1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["record", "reads"])
    ap.add_argument("--bases", type=int, default=248956422)
    ap.add_argument("--total", type=int, default=1_000_000_000)
    ap.add_argument("--M", type=int, default=500)
    ap.add_argument("--jobs", type=int, default=0)
    ap.add_argument("--workers", type=int, default=min(12, os.cpu_count() or 1))
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--dir", default="/tmp/m500")
    ap.add_argument("--stderr", default="", help="file that receives ribbit-hip's stderr")
    a = ap.parse_args()
    os.makedirs(a.dir, exist_ok=True)
    fasta, bed = os.path.join(a.dir, f"{a.mode}.fa"), os.path.join(a.dir, f"{a.mode}.bed")
    t0 = time.perf_counter()
    if a.mode == "record":
        make_record((a.bases, a.M, fasta))
        total, records = a.bases, 1
    else:
        piece = 50_000_000
        jobs = [(k, min(piece, a.total - k * piece), a.M, os.path.join(a.dir, f"piece{k:03d}.fa")) for k in range((a.total + piece - 1) // piece)]
        with ProcessPoolExecutor(max_workers=a.workers) as pool:
            made = list(pool.map(make_reads, jobs))
        records = sum(k for _, k in made)
        with open(fasta, "wb") as out:
            for p, _ in made:
                with open(p, "rb") as fh:
                    while True:
                        chunk = fh.read(64 << 20)
                        if not chunk:
                            break
                        out.write(chunk)
                os.remove(p)
        total = a.total
    t_gen = time.perf_counter() - t0
    cmd = [os.path.join(ROOT, "ribbit_amd", "ribbit-hip"), "-i", fasta, "-o", bed, "-m", "2", "-M", str(a.M)]
    if a.jobs:
        cmd += ["--jobs", str(a.jobs)]
    print(f"generated {records} record(s), {total} bases in {t_gen:.1f} s; running {' '.join(cmd)}", file=sys.stderr, flush=True)
    vram = VramPeak()
    vram.t.start()
    errpath = a.stderr or os.path.join(a.dir, f"{a.mode}.stderr")
    t1 = time.perf_counter()
    with open(errpath, "w") as errf:
        r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=errf, env=dict(os.environ, RIBBIT_PROFILE="1"))
    wall = time.perf_counter() - t1
    vram.stop = True
    rss_kb = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    err = open(errpath).read()
    if r.returncode != 0:
        print(err[-4000:], file=sys.stderr)
        print(json.dumps({"workload": a.mode, "M": a.M, "bases": total, "rc": r.returncode, "wall_s": wall, "error_tail": err[-600:]}), flush=True)
        raise SystemExit(r.returncode)
    rows = sum(1 for _ in open(bed, "rb"))
    tail = [l for l in err.splitlines() if l.startswith("[stages") or l.startswith("[devices]") or l.startswith("[refine_bed] cumulative")]
    print(json.dumps({"workload": f"{a.mode}: {records} record(s), -m 2 -M {a.M}", "bases": total, "records": records, "wall_s": round(wall, 3),
                      "mbases_per_s": round(total / wall / 1e6, 3), "records_per_s": round(records / wall, 2), "bed_rows": rows,
                      **vram.report(err), "peak_host_rss_gb": round(rss_kb / 1e6, 2),
                      "generate_s": round(t_gen, 1), "jobs": a.jobs or "auto", "profile": tail[-3:]}), flush=True)
    if not a.keep:
        os.remove(fasta); os.remove(bed)


if __name__ == "__main__":
    main()
