#!/usr/bin/env python3
"""The anchored merge's two first passes against each other at a size the oracle cannot reach in a GPU box's time limit: the seed
lists, the dispatch order and the SHA-256 of the BED text of one record with the pass on the GPU's lanes (the default from 2^20 kept
calls) and on the host threads alone (RIBBIT_DEVICE_MERGE_MIN set beyond any record).  A child process per form (the threshold is
read when the stage runs; a fresh process keeps the two honest).  usage: merge_paths_agree.py BASES [MAX_MOTIF]"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import hashlib, json, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
import ribbit_amd
from ribbit_amd import simulate
bases, m_hi = %(bases)d, %(m_hi)d
seq = simulate.m500_record(bases) if m_hi > 100 else simulate.grch38_shaped_record(1, bases, 2, m_hi)
with ribbit_amd.Scanner(2, m_hi) as sc:
    sc.load_record(seq)
    t = time.time()
    p, s, a = sc.processShiftXORsAnchored(copy=False)
    d = sc.dispatch_seeds(copy=False)
    t1 = time.time()
    digest = {k: hashlib.sha256(np.ascontiguousarray(v).view(np.uint8)).hexdigest() for k, v in (("perfect", p), ("subst", s), ("anchored", a), ("dispatch", d))}
    stats = ribbit_amd.last_device_merge()
    bed = sc.refine_bed_view("x")
    t2 = time.time()
    digest["bed"] = hashlib.sha256(np.ascontiguousarray(bed)).hexdigest()
    print(json.dumps({"digests": digest, "rows": int((bed == 10).sum()), "anchored_seeds": int(len(a)), "device_merge": stats,
                      "scans_and_merges_s": t1 - t, "refinement_s": t2 - t1}))
"""
bases = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
m_hi = int(sys.argv[2]) if len(sys.argv) > 2 else 100
out = {}
for form, env in (("device", {}), ("host", {"RIBBIT_DEVICE_MERGE_MIN": "4000000000"})):
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "bases": bases, "m_hi": m_hi}], env=dict(os.environ, **env), capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit(f"{form}: {r.stderr[-2000:]}")
    out[form] = json.loads(r.stdout.strip().splitlines()[-1])
same = out["device"]["digests"] == out["host"]["digests"]
print(json.dumps({"bases": bases, "max_motif": m_hi, "identical": same, "device": out["device"], "host": out["host"]}))
sys.exit(0 if same and out["device"]["device_merge"][0] > 0 and out["host"]["device_merge"][0] == 0 else 1)
