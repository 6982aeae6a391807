#!/usr/bin/env python3
"""Full-size identity for the driver's records: the ORACLE pipeline (oracle/, CPU, one core per
record) on the records of the GRCh38-shaped set -- record 0 is exactly the chromosome-1-sized record of bench.py's
`chr1_full_path` leg -- and, per record, the SHA-256 and row count of the BED text it writes.

    python tests/golden/make_full_size_digests.py [--records 0,1,...|all] [--budget-gb 40] [--scale 1.0] [--out tests/golden/grch38_shape_digests.json]

Run in the BUILD container (no GPU): ~4 s and ~70 MB per Mbp on one core (the oracle's planes are bits since round 4; with
byte planes a chromosome needed 62 GB).  Records run side by side as far as --budget-gb allows, largest first; the JSON is
rewritten after every record, so an interrupted run keeps what it has and a later run only adds what is missing.
bench.py (`chr1_full_path.verified`) and tools/grch38_shape.py (`verified_records`) compare the GPU path's BED against these
digests outside their timed regions.  The digests lock the ORACLE's output (parity unpinned), nothing more."""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GB_PER_MBP = 0.075


def one_record(k: int, scale: float):
    from oracle_lib import Oracle
    from ribbit_amd.simulate import GRCH38_RECORDS, grch38_shaped_record
    name, full = GRCH38_RECORDS[k]
    bases = max(1000, int(full * scale))
    seq = grch38_shaped_record(k, bases)
    t0 = time.time()
    with Oracle(seq, 2, 100) as o:
        o.run_all()
        bed = o.refine_bed_bytes(name)
    return {"record": k, "name": name, "bases": bases, "generator_seed": 4 + 1000 * k, "m_lo": 2, "m_hi": 100,
            "bed_rows": bed.count(b"\n"), "bed_bytes": len(bed), "sha256": hashlib.sha256(bed).hexdigest(), "oracle_seconds": round(time.time() - t0, 1)}


def m500_digest(bases: int, out: str):
    """the oracle pipeline on ribbit_amd.simulate.m500_record(bases) at -m 2 -M 500 (~25 s per Mbp on one core here) -> entry
    "m500" of the digest file: what bench.py's `m500_full_path` leg checks its BED against"""
    from oracle_lib import Oracle
    from ribbit_amd.simulate import m500_record
    seq = m500_record(bases)
    t0 = time.time()
    with Oracle(seq, 2, 500) as o:
        o.run_all()
        bed = o.refine_bed_bytes("m500")
    have = json.load(open(out)) if os.path.exists(out) else {}
    have["m500"] = {"bases": bases, "generator_seed": 77, "m_lo": 2, "m_hi": 500, "bed_rows": bed.count(b"\n"), "bed_bytes": len(bed),
                    "sha256": hashlib.sha256(bed).hexdigest(), "oracle_seconds": round(time.time() - t0, 1), "sequence_id": "m500"}
    json.dump(have, open(out + ".tmp", "w"), indent=1, sort_keys=True)
    os.replace(out + ".tmp", out)
    print(have["m500"], flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m500", type=int, default=0, help="instead of the set: the -M 500 record of this many bases (bench.py uses 16000000)")
    ap.add_argument("--records", default="all")
    ap.add_argument("--budget-gb", type=float, default=40.0)
    ap.add_argument("--max-workers", type=int, default=max(1, (os.cpu_count() or 2) - 2))
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "grch38_shape_digests.json"))
    ap.add_argument("--one", type=int, default=-1, help=argparse.SUPPRESS)       # worker mode: one record, JSON on stdout
    a = ap.parse_args()
    if a.m500:
        m500_digest(a.m500, a.out)
        return
    if a.one >= 0:
        print(json.dumps(one_record(a.one, a.scale)), flush=True)
        return
    from ribbit_amd.simulate import GRCH38_RECORDS
    have = {}
    if os.path.exists(a.out):
        have = json.load(open(a.out))
    key = "records" if a.scale == 1.0 else f"records_scale_{a.scale}"
    done = have.setdefault(key, {})
    want = list(range(len(GRCH38_RECORDS))) if a.records == "all" else [int(x) for x in a.records.split(",")]
    todo = sorted((k for k in want if GRCH38_RECORDS[k][0] not in done), key=lambda k: -GRCH38_RECORDS[k][1])
    running = {}        # k -> (process, GB)
    t0 = time.time()

    def gb(k):
        return GRCH38_RECORDS[k][1] * a.scale / 1e6 * GB_PER_MBP + 0.5

    while todo or running:
        used = sum(g for _, g in running.values())
        for k in list(todo):
            if len(running) < a.max_workers and (used + gb(k) <= a.budget_gb or not running):
                p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--one", str(k), "--scale", str(a.scale)], stdout=subprocess.PIPE, text=True)
                running[k] = (p, gb(k)); used += gb(k); todo.remove(k)
        time.sleep(5)
        for k, (p, _) in list(running.items()):
            if p.poll() is None:
                continue
            out = p.stdout.read()
            del running[k]
            if p.returncode != 0:
                print(f"record {k} failed (rc {p.returncode})", file=sys.stderr, flush=True)
                continue
            r = json.loads(out.strip().splitlines()[-1])
            done[r["name"]] = r
            have["what"] = ("SHA-256 and row count of the BED text the ORACLE pipeline (oracle/, CPU) writes for the records of "
                            "ribbit_amd.simulate.grch38_shaped_record, -m 2 -M 100; made by tests/golden/make_full_size_digests.py")
            tmp = a.out + ".tmp"
            json.dump(have, open(tmp, "w"), indent=1, sort_keys=True)
            os.replace(tmp, a.out)
            print(f"[{time.time() - t0:7.0f} s] {r['name']}: {r['bases']} bases, {r['bed_rows']} rows, {r['oracle_seconds']} s, {r['sha256'][:16]}", flush=True)


if __name__ == "__main__":
    main()
