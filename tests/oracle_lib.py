"""ctypes access to the CPU oracle (oracle/libribbit_oracle.so).  TEST-ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")

RANK = {"P": 5, "Q": 4, "S": 3, "F": 2, "C": 1, "A": 0, "N": -1}
LIST_PERFECT, LIST_SUBST, LIST_ANCHORED = 0, 1, 2

JOB_DT = np.dtype([("seed_index", "<i4"), ("seed_type", "<i4"), ("motif_length", "<i4"), ("atomicity", "<i4"),
                   ("query_start", "<i4"), ("query_length", "<i4"), ("ppr_length", "<i4"), ("small", "<i4"),
                   ("motif_offset", "<i4")])
SEED_DT = np.dtype([("start", "<i4"), ("end", "<i4"), ("mlen", "<i4"), ("type", "<i4")])
CALL_DT = np.dtype([("pos", "<i4"), ("mlen", "<i4"), ("start", "<i4"), ("end", "<i4")])


def build(asan: bool = False) -> str:
    name = "libribbit_oracle_asan.so" if asan else "libribbit_oracle.so"
    path = os.path.join(_ORACLE_DIR, name)
    srcs = [os.path.join(_ORACLE_DIR, f) for f in ("ribbit_oracle.c", "ribbit_oracle_refine.cpp", "ribbit_oracle.h", "Makefile")]
    if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR, path])
    return path


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.rbo_open.restype = C.c_void_p
        L.rbo_open.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int]
        L.rbo_close.argtypes = [C.c_void_p]
        for f in ("rbo_min_shift", "rbo_max_shift", "rbo_run_perfect", "rbo_run_subst",
                  "rbo_run_anchor_planes", "rbo_run_anchored", "rbo_run_dispatch"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [C.c_void_p]
        L.rbo_length.restype = C.c_int64
        L.rbo_length.argtypes = [C.c_void_p]
        L.rbo_guard_hits.restype = C.c_int64
        L.rbo_guard_hits.argtypes = [C.c_void_p]
        L.rbo_range_queries.restype = C.c_int64
        L.rbo_range_queries.argtypes = [C.c_void_p]
        for f in ("rbo_plane", "rbo_anchor_plane"):
            getattr(L, f).restype = C.POINTER(C.c_uint8)
            getattr(L, f).argtypes = [C.c_void_p, C.c_int]
        for f in ("rbo_nmask", "rbo_codes"):
            getattr(L, f).restype = C.POINTER(C.c_uint8)
            getattr(L, f).argtypes = [C.c_void_p]
        L.rbo_seeds.restype = C.c_int64
        L.rbo_seeds.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.rbo_calls.restype = C.c_int64
        L.rbo_calls.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.rbo_dispatch.restype = C.c_int64
        L.rbo_dispatch.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.rbo_refine_params_default.restype = None
        L.rbo_refine_params_default.argtypes = [C.POINTER(RefineParams), C.c_int, C.c_int]
        L.rbo_refine_jobs.restype = C.c_int64
        L.rbo_refine_jobs.argtypes = [C.c_void_p, C.POINTER(RefineParams), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.rbo_refine_bed.restype = C.c_void_p
        L.rbo_refine_bed.argtypes = [C.c_void_p, C.POINTER(RefineParams), C.c_char_p, C.c_char_p, C.POINTER(C.c_int64)]
        L.rbo_range_count.restype = C.c_int
        L.rbo_range_count.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        _lib = L
    return _lib


def _copy(ptr, n, dt):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dt)
    buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt).copy()


class RefineParams(C.Structure):
    _fields_ = [("min_length", C.c_int32 * 1024), ("perfect_units", C.c_int32 * 1024),
                ("purity_threshold", C.c_float), ("continuous_ones_threshold", C.c_int32)]


class Oracle:
    """Staged access to the restated processSequence (fasta_utils.cpp:59-250)."""

    def __init__(self, seq: bytes, m_lo: int = 2, m_hi: int = 100):
        self._L = lib()
        self.seq = bytes(seq)
        self.m_lo, self.m_hi = m_lo, m_hi
        self._h = self._L.rbo_open(self.seq, len(self.seq), m_lo, m_hi)
        self.length = len(self.seq)
        self.min_shift = self._L.rbo_min_shift(self._h)
        self.max_shift = self._L.rbo_max_shift(self._h)

    def close(self):
        if self._h:
            self._L.rbo_close(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _bytes(self, ptr):
        if not ptr:
            raise ValueError("plane not available")
        return np.ctypeslib.as_array(ptr, shape=(self.length,)).copy() if self.length else np.zeros(0, np.uint8)

    def plane(self, shift):
        return self._bytes(self._L.rbo_plane(self._h, shift))

    def anchor_plane(self, shift):
        return self._bytes(self._L.rbo_anchor_plane(self._h, shift))

    def nmask(self):
        return self._bytes(self._L.rbo_nmask(self._h))

    def codes(self):
        return self._bytes(self._L.rbo_codes(self._h))

    def run_perfect(self):
        return self._L.rbo_run_perfect(self._h)

    def run_subst(self):
        return self._L.rbo_run_subst(self._h)

    def run_anchor_planes(self):
        return self._L.rbo_run_anchor_planes(self._h)

    def run_anchored(self):
        return self._L.rbo_run_anchored(self._h)

    def run_dispatch(self):
        return self._L.rbo_run_dispatch(self._h)

    def run_all(self):
        self.run_perfect(); self.run_subst(); self.run_anchor_planes(); self.run_anchored()
        return self.run_dispatch()

    def seeds(self, which):
        p = C.c_void_p()
        n = self._L.rbo_seeds(self._h, which, C.byref(p))
        return _copy(p.value, n, SEED_DT)

    def calls(self, which):
        p = C.c_void_p()
        n = self._L.rbo_calls(self._h, which, C.byref(p))
        return _copy(p.value, n, CALL_DT)

    def dispatch(self):
        p = C.c_void_p()
        n = self._L.rbo_dispatch(self._h, C.byref(p))
        return _copy(p.value, n, SEED_DT)

    def refine_jobs(self, params=None):
        """(jobs array, motif pool bytes); needs run_dispatch() first"""
        rp = params
        if rp is None:
            rp = RefineParams()
            self._L.rbo_refine_params_default(C.byref(rp), self.m_lo, self.m_hi)
        jobs, pool = C.c_void_p(), C.c_void_p()
        n = self._L.rbo_refine_jobs(self._h, C.byref(rp), C.byref(jobs), C.byref(pool))
        arr = _copy(jobs.value, n, JOB_DT)
        size = int((arr["motif_offset"] + arr["atomicity"]).max()) if len(arr) else 0
        return arr, (C.string_at(pool.value, size) if size else b"")

    def refine_bed(self, seq_id: str = "seq", params=None) -> str:
        """BED text of this record (needs run_dispatch()); alignments by the reference's own SSW (oracle/_ref)"""
        rp = params
        if rp is None:
            rp = RefineParams()
            self._L.rbo_refine_params_default(C.byref(rp), self.m_lo, self.m_hi)
        return self.refine_bed_bytes(seq_id, rp).decode()

    def refine_bed_bytes(self, seq_id: str = "seq", params=None) -> bytes:
        """the same text as bytes (full-size records: 150 MB, hashed rather than compared as a str)"""
        rp = params
        if rp is None:
            rp = RefineParams()
            self._L.rbo_refine_params_default(C.byref(rp), self.m_lo, self.m_hi)
        n = C.c_int64()
        p = self._L.rbo_refine_bed(self._h, C.byref(rp), self.seq, seq_id.encode(), C.byref(n))
        return C.string_at(p, n.value)

    def range_count(self, shift, start, end):
        return self._L.rbo_range_count(self._h, shift, start, end)

    def guard_hits(self):
        return self._L.rbo_guard_hits(self._h)

    def range_queries(self):
        return self._L.rbo_range_queries(self._h)
