"""CPU checks of the drop-in boundary: libribbit_hip.so loads and exports every symbol that
include/ribbit_hip.h declares, and the product fails loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re

import pytest

import ribbit_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ribbit_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ribbit_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(hip_lib):
    declared = _declared_symbols()
    assert declared, "no declarations found in include/ribbit_hip.h"
    for name in declared:
        assert hasattr(hip_lib, name), f"{name} declared in ribbit_hip.h but not exported"
    assert sorted(ribbit_amd.ABI_SYMBOLS) == declared


def test_abi_version_and_defaults(hip_lib):
    assert hip_lib.ribbit_hip_abi_version() == 1
    p = ribbit_amd.ScanParams()
    hip_lib.ribbit_scan_params_default(C.byref(p), 2, 100)
    assert (p.min_motif, p.max_motif, p.window_length, p.subst_threshold, p.anchor_threshold, p.anchor_length) == (2, 100, 8, 7, 6, 3)


def test_no_oracle_dependency_in_product():
    # the product must never link, import or call oracle/ code
    # (tools/ is not product either, but it is not test infrastructure: the scripts there must not use the oracle;
    #  the checkers that do live under tests/sweeps/)
    for top in ("ribbit_amd", "tools"):
      for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
        for f in files:
            if f.endswith((".py", ".sh", ".cpp", ".hip", ".h", "Makefile")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in src and "ribbit_oracle" not in src and "rbo_" not in src, os.path.join(dirpath, f)
    import subprocess
    needed = subprocess.check_output(["readelf", "-d", ribbit_amd.library_path()]).decode()
    assert "oracle" not in needed


def test_open_fails_loudly_without_gpu(hip_lib):
    if hip_lib.ribbit_hip_device_count() > 0:
        pytest.skip("a gfx950 device is present")
    with pytest.raises(ribbit_amd.RibbitHipError):
        ribbit_amd.Scanner(2, 6)


def test_bad_arguments_rejected(hip_lib):
    p = ribbit_amd.ScanParams()
    hip_lib.ribbit_scan_params_default(C.byref(p), 5, 3)
    h = C.c_void_p()
    assert hip_lib.ribbit_hip_open(C.byref(p), 0, C.byref(h)) == -1
    assert b"motif range" in hip_lib.ribbit_hip_last_error()
    assert hip_lib.ribbit_hip_open(None, 0, C.byref(h)) == -1
