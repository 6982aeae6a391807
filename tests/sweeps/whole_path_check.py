"""One-off scale check on the GPU box: the whole path on a simulated record against the oracle pipeline.
Usage: python tests/sweeps/whole_path_check.py <bases> [seed]   (the oracle needs ~10 s per Mbp)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ribbit_amd
from oracle_lib import Oracle
from ribbit_amd.simulate import simulate_sequence

import threading


def _heartbeat():
    t0 = time.time()
    while True:
        time.sleep(60)
        print(f"... {time.time() - t0:.0f} s", flush=True)


threading.Thread(target=_heartbeat, daemon=True).start()
bases = int(sys.argv[1])
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
seq, _ = simulate_sequence(bases, seed, 2, 100, n_block_rate=0.1, lower_rate=0.1)
t = time.time()
with ribbit_amd.Scanner(2, 100) as sc:
    sc.load_record(seq)
    got = sc.refine_bed("chr")
print(f"GPU path {time.time() - t:.1f} s, {got.count(chr(10))} rows", flush=True)
t = time.time()
with Oracle(seq, 2, 100) as o:
    o.run_all()
    want = o.refine_bed("chr")
print(f"oracle {time.time() - t:.1f} s, {want.count(chr(10))} rows", flush=True)
print("IDENTICAL" if got == want else "DIFFERENT")
