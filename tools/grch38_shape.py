#!/usr/bin/env python3
"""BASELINE.json configs[2] / configs[3] as stated: the whole GRCh38-shaped set -- 24 records with the primary assembly's
lengths (chr1 248,956,422 ... chrY 57,227,415; 3.1 Gbp), generator output with N blocks at the record ends and one 3-Mbp
centromere-like N block per record (GRCh38 itself is not on the box) -- through the command-line front end,
`-m 2 -M 100`, records in flight, optionally dealt over several GPUs.

    python tools/grch38_shape.py [--scale 1.0] [--devices 0,1,...] [--jobs N] [--keep]

Writes the FASTA to /tmp (generation runs in a process pool and is outside the timed region), times `ribbit-hip`, prints
one JSON line: wall time, Mbases/s, BED rows.  --scale shrinks every record (0.1: a 310-Mbp set)."""
import argparse
import json
import os
import subprocess
import sys
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ribbit_amd.simulate import GRCH38_RECORDS as GRCH38


def make_record(args):
    """-> path of a file holding the record's FASTA text (header + 80-column lines)"""
    import numpy as np
    from ribbit_amd.simulate import grch38_shaped_record
    k, name, bases, out_dir = args
    b = np.frombuffer(grch38_shaped_record(k, bases), dtype=np.uint8)
    # 80-column lines without a Python loop over the lines
    full = bases // 80 * 80
    body = np.empty((full // 80, 81), dtype=np.uint8)
    body[:, :80] = b[:full].reshape(-1, 80)
    body[:, 80] = ord("\n")
    path = os.path.join(out_dir, f"rec{k:02d}.fa")
    with open(path, "wb") as fh:
        fh.write(b">" + name.encode() + b"\n")
        fh.write(body.tobytes())
        if full < bases:
            fh.write(b[full:].tobytes() + b"\n")
    return path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--devices", default="")
    ap.add_argument("--jobs", type=int, default=0)
    ap.add_argument("--workers", type=int, default=min(12, os.cpu_count() or 1))
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--records", type=int, default=len(GRCH38), help="only the first N records")
    ap.add_argument("--stderr", default="", help="file that receives ribbit-hip's stderr (progress and RIBBIT_PROFILE lines)")
    ap.add_argument("--dir", default="/tmp/grch38_shape")
    a = ap.parse_args()
    os.makedirs(a.dir, exist_ok=True)
    recs = [(k, name, max(1000, int(n * a.scale)), a.dir) for k, (name, n) in enumerate(GRCH38[:a.records])]
    t0 = time.perf_counter()
    with ProcessPoolExecutor(max_workers=a.workers) as pool:
        paths = list(pool.map(make_record, recs))
    fasta, bed = os.path.join(a.dir, "set.fa"), os.path.join(a.dir, "set.bed")
    with open(fasta, "wb") as out:
        for p in paths:
            with open(p, "rb") as fh:
                while True:
                    chunk = fh.read(64 << 20)
                    if not chunk:
                        break
                    out.write(chunk)
            os.remove(p)
    t_gen = time.perf_counter() - t0
    total = sum(r[2] for r in recs)
    cmd = [os.path.join(ROOT, "ribbit_amd", "ribbit-hip"), "-i", fasta, "-o", bed, "-m", "2", "-M", "100"]
    if a.devices:
        cmd += ["--devices", a.devices]
    if a.jobs:
        cmd += ["--jobs", str(a.jobs)]
    print(f"generated {len(recs)} records, {total} bases in {t_gen:.1f} s; running {' '.join(cmd)}", file=sys.stderr, flush=True)
    t1 = time.perf_counter()
    r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=dict(os.environ, RIBBIT_PROFILE="1"))
    wall = time.perf_counter() - t1
    if a.stderr:
        open(a.stderr, "w").write(r.stderr)
    if r.returncode != 0:
        print(r.stderr[-3000:], file=sys.stderr)
        raise SystemExit(r.returncode)
    # every record's rows against the digest the oracle pipeline wrote for it (tests/golden/grch38_shape_digests.json, made on
    # the CPU by tests/golden/make_full_size_digests.py): SHA-256 and row count per record, outside the timed region
    import hashlib
    rows, per_record = 0, {}
    cur, h, n = None, None, 0
    with open(bed, "rb") as fh:
        for line in fh:
            name = line[:line.index(b"\t")].decode()
            if name != cur:
                if cur is not None:
                    per_record[cur] = (h.hexdigest(), n)
                cur, h, n = name, hashlib.sha256(), 0
            h.update(line); n += 1; rows += 1
    if cur is not None:
        per_record[cur] = (h.hexdigest(), n)
    digests = {}
    try:
        allv = json.load(open(os.path.join(ROOT, "tests", "golden", "grch38_shape_digests.json")))
        digests = allv.get("records" if a.scale == 1.0 else f"records_scale_{a.scale}", {})
    except (OSError, ValueError):
        pass
    checked = {name: (per_record.get(name, ("", 0)) == (d["sha256"], d["bed_rows"])) for (_, name, bases, _) in recs
               for d in [digests.get(name)] if d and d["bases"] == bases}
    verified = {"records_checked": len(checked), "records_identical": sum(checked.values()), "records_without_digest": len(recs) - len(checked),
                "different": sorted(k for k, v in checked.items() if not v),
                "what": "per record: SHA-256 and row count of its BED rows == the CPU oracle pipeline's (tests/golden/grch38_shape_digests.json)"}
    tail = [l for l in r.stderr.splitlines() if l.startswith("[stages") or l.startswith("[devices]") or l.startswith("[shared")]
    print(json.dumps({"workload": f"GRCh38-shaped set, {len(recs)} records, scale {a.scale}", "bases": total, "wall_s": wall, "mbases_per_s": total / wall / 1e6,
                      "bed_rows": rows, "verified_records": verified, "devices": a.devices or "0", "generate_s": t_gen, "profile": tail}), flush=True)
    if not a.keep:
        os.remove(fasta); os.remove(bed)


if __name__ == "__main__":
    main()
