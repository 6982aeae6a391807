"""One HIP runtime per process whichever of torch and libribbit_hip.so comes first (ribbit_amd._share_torchs_hip_runtime):
both import orders in child processes; in each torch must see the GPU and alias the library's device memory as a tensor --
what the RCCL exchange of the chunk-sharded path does with the runs a scan leaves on the device."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_torch_sees_the_gpu_in_either_import_order():
    pytest.importorskip("torch")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "torch_order_probe.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 2 and all("torch sees the GPU" in l for l in lines), r.stdout + r.stderr[-1500:]
