import os
import sys

import pytest

# torch BEFORE anything loads libribbit_hip.so.  The torch wheel bundles its own HIP and HSA runtimes (torch/lib/libamdhip64.so,
# SONAME libamdhip64.so.7, found through $ORIGIN); the library needs libamdhip64.so.7 from /opt/rocm.  With torch first the
# library's NEEDED entry is satisfied by the copy already in the process (same SONAME): one runtime.  The other way round torch
# asks for "libamdhip64.so", which is not the name of anything loaded, finds its own file and brings a SECOND runtime into a
# process whose first one holds the device: torch then reports "No HIP GPUs are available" (readelf -d on the three files;
# tools/torch_order_probe.py shows both orders).  It made test_sharded_gpu.py fail as a file while each of its tests passed
# alone.  bench.py imports torch at its top for the same reason; with torch first the library runs on torch's bundled runtime.
# Since round 4 the order no longer matters -- ribbit_amd.load_library() loads torch's bundled runtime itself when a torch is
# installed (tests/test_torch_order_gpu.py) -- and the import below only saves the suite a second of start-up later.
try:
    import torch  # noqa: F401
except ImportError:            # the CPU-only parts of the suite do not need it
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def hip_lib():
    import ribbit_amd
    return ribbit_amd.load_library()
