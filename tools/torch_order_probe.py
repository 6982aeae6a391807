#!/usr/bin/env python3
"""torch imported before / after libribbit_hip.so is loaded: both orders in child processes on the GPU box, and what torch
sees in each.  Until round 4 the second order put two HIP runtimes into the process (tests/conftest.py has the mechanism)
and torch reported "No HIP GPUs are available"; ribbit_amd.load_library() now loads torch's bundled runtime first when a
torch is installed, so both orders must print "torch sees the GPU" (tests/test_torch_order_gpu.py runs this)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %r)
order = sys.argv[1]
if order == "torch_first":
    import torch
import ribbit_amd
with ribbit_amd.Scanner(2, 20) as sc:
    sc.load_record(b"ACGT" * 1000)
    sc.scan_perfect_begin()
    ptr, n, hptr, nh = sc.scan_perfect_end_device()
    import torch
    from ribbit_amd.distributed import device_bytes
    try:
        t = device_bytes(ptr, n * 16, torch.device("cuda", 0))
        print(order, ": torch sees the GPU,", n, "run records aliased as a tensor of", t.numel(), "bytes")
    except RuntimeError as e:
        print(order, ": FAILED:", str(e)[:200])
''' % ROOT

for order in ("torch_first", "library_first"):
    r = subprocess.run([sys.executable, "-c", CHILD, order], capture_output=True, text=True, timeout=300)
    print((r.stdout.strip() or r.stderr.strip()[-300:]))
