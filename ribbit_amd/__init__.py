"""ribbit_amd -- MI355X-native implementation of ribbit's shift-XOR tandem-repeat scan.

This module is only the Python face of the C ABI in include/ribbit_hip.h (ctypes, no torch
types cross the boundary).  All compute happens in ribbit_amd/libribbit_hip.so (hand-written
gfx950 kernels + C++ host logic).  There is no CPU fallback: importing works anywhere, but
opening a scanner without the built library or without a gfx950 GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

__all__ = ["RibbitHipError", "ScanParams", "Scanner", "library_path", "load_library", "host_replay_calls", "pack_planes", "pack_bit_planes",
           "RUN_DT", "CALL_DT", "SEED_DT", "JOB_DT", "ENDS_DT", "RANK", "TERM", "RefineParams", "host_refine_jobs", "host_refine_bed", "host_merge_chunks", "host_perfect_runs_from_events", "pair_halves", "ssw_align", "ssw_align_periodic", "merge_chunk_runs", "join_run_halves",
           "RUN_NOT_OWNED", "RUN_HALF_START", "RUN_HALF_END"]

_HERE = os.path.dirname(os.path.abspath(__file__))

RANK = {"P": 5, "Q": 4, "S": 3, "F": 2, "C": 1, "A": 0, "N": -1}     # global_variables.cpp:28-34
TERM = {"ZERO": 0, "N": 1, "EOS": 2}

JOB_DT = np.dtype([("seed_index", "<i4"), ("seed_type", "<i4"), ("motif_length", "<i4"), ("atomicity", "<i4"),
                   ("query_start", "<i4"), ("query_length", "<i4"), ("ppr_length", "<i4"), ("small", "<i4"),
                   ("motif_offset", "<i4")])
ENDS_DT = np.dtype([(n, "<i4") for n in ("score", "ref_end", "query_end", "score2", "ref_end2", "ref_begin", "query_begin", "flag")])
RUN_DT = np.dtype([("start", "<i4"), ("end", "<i4"), ("mlen", "<i4"), ("term", "<i4")])
CALL_DT = np.dtype([("pos", "<i4"), ("mlen", "<i4"), ("start", "<i4"), ("end", "<i4")])
SEED_DT = np.dtype([("start", "<i4"), ("end", "<i4"), ("mlen", "<i4"), ("type", "<i4")])

# every symbol include/ribbit_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "ribbit_scan_params_default", "ribbit_hip_last_error", "ribbit_hip_abi_version",
    "ribbit_hip_device_count", "ribbit_hip_open", "ribbit_hip_close", "ribbit_hip_set_stream",
    "ribbit_hip_load_record", "ribbit_hip_load_record_device", "ribbit_hip_load_record_pinned", "ribbit_hip_host_alloc", "ribbit_hip_host_free",
    "ribbit_fasta_open", "ribbit_fasta_next", "ribbit_fasta_release", "ribbit_fasta_close", "ribbit_fasta_last_error", "ribbit_hip_scan_perfect_runs",
    "ribbit_hip_perfect_calls", "ribbit_hip_seeds_perfect", "ribbit_hip_plane_bits",
    "ribbit_hip_range_popcount", "ribbit_hip_plane_words", "ribbit_hip_packed_plane",
    "ribbit_hip_last_timing_ms", "ribbit_hip_last_event_count",
    "ribbit_hip_subst_calls", "ribbit_hip_seeds_substitutions",
    "ribbit_host_replay_calls", "ribbit_seed_lists_free", "ribbit_host_longest_runs", "ribbit_debug_set_merge_min_range", "ribbit_debug_last_merge",
    "ribbit_hip_small_motifs", "ribbit_debug_small_motif_counters", "ribbit_debug_last_dispatch_ranges", "ribbit_debug_last_device_merge", "ribbit_debug_alignment_counters", "ribbit_debug_level_counters",
    "ribbit_hip_adopt_dispatch", "ribbit_hip_refine_met_empty_query", "ribbit_hip_device_pci_bus_id",
    "ribbit_hip_anchored_calls", "ribbit_hip_seeds_anchored", "ribbit_hip_dispatch_seeds", "ribbit_hip_guard_hits",
    "ribbit_hip_debug_stream_read",
    "ribbit_refine_params_default", "ribbit_hip_seed_longest_runs", "ribbit_hip_refine_jobs",
    "ribbit_host_refine_jobs", "ribbit_refine_jobs_free", "ribbit_ssw_align", "ribbit_debug_ssw_align_periodic",
    "ribbit_hip_refine_bed", "ribbit_host_refine_bed", "ribbit_text_free",
    "ribbit_hip_xa_words",
    "ribbit_host_perfect_runs_from_events", "ribbit_runs_free", "ribbit_hip_perfect_runs_partial",
    "ribbit_hip_scan_perfect_chunk", "ribbit_hip_host_register", "ribbit_hip_host_unregister",
    "ribbit_hip_set_host_threads", "ribbit_hip_ssw_passes", "ribbit_hip_ssw_align_jobs", "ribbit_hip_set_timing", "ribbit_hip_debug_set_event_capacity", "ribbit_hip_debug_pair_events", "ribbit_hip_scan_perfect_begin", "ribbit_hip_scan_perfect_end", "ribbit_hip_scan_perfect_wait", "ribbit_hip_scan_perfect_end_device",
    "ribbit_hip_stage_calls_chunk", "ribbit_hip_xa_words_strided", "ribbit_host_merge_chunks",
]


class RibbitHipError(RuntimeError):
    pass


class ScanParams(C.Structure):
    """RibbitScanParams (ribbit.cpp:191,240-243; fasta_utils.cpp:165)."""
    _fields_ = [("min_motif", C.c_int32), ("max_motif", C.c_int32), ("window_length", C.c_int32),
                ("subst_threshold", C.c_int32), ("anchor_threshold", C.c_int32), ("anchor_length", C.c_int32)]


class RefineParams(C.Structure):
    """RibbitRefineParams (MINIMUM_LENGTH / PERFECT_UNITS tables, purity, cones threshold)."""
    _fields_ = [("min_length", C.c_int32 * 1024), ("perfect_units", C.c_int32 * 1024),
                ("purity_threshold", C.c_float), ("continuous_ones_threshold", C.c_int32)]


class Alignment(C.Structure):
    """RibbitAlignment (StripedSmithWaterman::Alignment without the strings)."""
    _fields_ = [(n, C.c_int32) for n in ("sw_score", "sw_score_next_best", "ref_begin", "ref_end", "query_begin",
                                         "query_end", "ref_end_next_best", "mismatches", "flag", "cigar_len")]


class SeedLists(C.Structure):
    """RibbitSeedLists (malloc'ed arrays returned by ribbit_host_replay_calls)."""
    _fields_ = [("perfect", C.c_void_p), ("n_perfect", C.c_size_t), ("subst", C.c_void_p), ("n_subst", C.c_size_t),
                ("anchored", C.c_void_p), ("n_anchored", C.c_size_t), ("dispatch", C.c_void_p), ("n_dispatch", C.c_size_t),
                ("guard_hits", C.c_int64)]


class ChunkCalls(C.Structure):
    """RibbitChunkCalls: what one window stage of one chunk of a longer record keeps (ribbit_hip_stage_calls_chunk)."""
    _fields_ = [("calls", C.c_void_p), ("n", C.c_size_t), ("pend", C.c_void_p), ("tail_pend", C.c_int32), ("inexact", C.c_int32),
                ("flush", C.c_void_p), ("n_flush", C.c_size_t), ("dev_calls", C.c_void_p), ("dev_pend", C.c_void_p), ("streaks", C.c_int64)]


class ChunkPart(C.Structure):
    """RibbitChunkPart: one chunk's contribution to ribbit_host_merge_chunks."""
    _fields_ = [("runs", C.c_void_p), ("n_runs", C.c_size_t), ("halves", C.c_void_p), ("n_halves", C.c_size_t),
                ("subst", ChunkCalls), ("anchored", ChunkCalls)]


STAGE_PERFECT, STAGE_SUBST, STAGE_ANCHORED = 0, 1, 2


def library_path() -> str:
    # RIBBIT_HIP_LIBRARY: another build of the same library (the sanitizer build of the CPU tests)
    return os.environ.get("RIBBIT_HIP_LIBRARY") or os.path.join(_HERE, "libribbit_hip.so")


_lib = None
loaded_before_torch = False      # libribbit_hip.so entered the process before torch did: torch.cuda will not work in it


def _share_torchs_hip_runtime() -> bool:
    """One HIP runtime per process, whichever of torch and this library comes first (round 4; until then the multi-GPU bench
    leg depended on `import torch` coming first).  The torch wheel bundles its own runtime (torch/lib/libamdhip64.so, SONAME
    libamdhip64.so.7) and asks for it by FILE name; libribbit_hip.so needs "libamdhip64.so.7" and would otherwise pull in
    /opt/rocm's copy -- two runtimes, and the second one (torch's) finds the device taken.  If a torch is installed and not yet
    imported, its bundled runtime is loaded HERE first (by path, globally): the library's NEEDED entry is then satisfied by it
    (same SONAME), and a later `import torch` finds the very file it asks for already in the process.  Without torch installed
    nothing happens and the library runs on /opt/rocm's runtime, as the command-line front end does.  torch itself is not
    imported.  -> True if torch's runtime is now the process's (or already was)."""
    if "torch" in sys.modules:
        return True
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return False
    if spec is None or not spec.submodule_search_locations:
        return False
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(cand):
        return False
    try:
        C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except OSError:
        return False
    return True


def load_library():
    """Load libribbit_hip.so; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RibbitHipError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(make -C ribbit_amd/csrc). ribbit_amd has no CPU fallback.")
    global loaded_before_torch
    loaded_before_torch = "torch" not in sys.modules and not _share_torchs_hip_runtime()      # see ribbit_amd.distributed.device_bytes
    L = C.CDLL(path)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.ribbit_scan_params_default.restype = None
    L.ribbit_scan_params_default.argtypes = [C.POINTER(ScanParams), i32, i32]
    L.ribbit_hip_last_error.restype = C.c_char_p
    L.ribbit_hip_last_error.argtypes = []
    L.ribbit_hip_abi_version.restype = C.c_int
    L.ribbit_hip_device_count.restype = C.c_int
    L.ribbit_hip_open.argtypes = [C.POINTER(ScanParams), C.c_int, C.POINTER(vp)]
    L.ribbit_hip_close.argtypes = [vp]
    L.ribbit_hip_set_stream.argtypes = [vp, vp]
    L.ribbit_hip_load_record.argtypes = [vp, C.c_char_p, i64]
    L.ribbit_hip_load_record_device.argtypes = [vp, vp, i64]
    L.ribbit_hip_load_record_pinned.argtypes = [vp, vp, i64]
    L.ribbit_hip_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.ribbit_hip_host_free.argtypes = [vp]
    L.ribbit_fasta_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.ribbit_fasta_next.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(vp), C.POINTER(i64), C.POINTER(C.c_int)]
    L.ribbit_fasta_release.argtypes = [vp, vp]
    L.ribbit_fasta_close.argtypes = [vp]
    L.ribbit_fasta_last_error.restype = C.c_char_p
    for f in ("ribbit_hip_scan_perfect_runs", "ribbit_hip_perfect_calls", "ribbit_hip_seeds_perfect", "ribbit_hip_subst_calls",
              "ribbit_hip_anchored_calls", "ribbit_hip_dispatch_seeds"):
        getattr(L, f).argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ribbit_hip_seeds_substitutions.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ribbit_hip_seeds_anchored.argtypes = [vp] + [C.POINTER(vp), C.POINTER(C.c_size_t)] * 3
    L.ribbit_hip_guard_hits.restype = i64
    L.ribbit_hip_guard_hits.argtypes = [vp]
    L.ribbit_host_replay_calls.argtypes = [C.POINTER(ScanParams), i64, vp, vp, vp, C.c_size_t, vp, C.c_size_t,
                                           vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(SeedLists)]
    L.ribbit_seed_lists_free.restype = None
    L.ribbit_seed_lists_free.argtypes = [C.POINTER(SeedLists)]
    L.ribbit_debug_last_merge.restype = None
    L.ribbit_debug_last_merge.argtypes = [C.c_int, C.POINTER(C.c_int32 * 5)]
    L.ribbit_debug_last_device_merge.restype = None
    L.ribbit_debug_last_device_merge.argtypes = [C.POINTER(C.c_int32 * 5)]
    L.ribbit_debug_last_dispatch_ranges.restype = C.c_int32
    L.ribbit_debug_last_dispatch_ranges.argtypes = []
    L.ribbit_debug_alignment_counters.restype = None
    L.ribbit_debug_alignment_counters.argtypes = [C.POINTER(C.c_int64 * 3)]
    L.ribbit_hip_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    L.ribbit_hip_adopt_dispatch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.ribbit_hip_refine_met_empty_query.argtypes = [C.c_void_p]
    L.ribbit_debug_level_counters.restype = None
    L.ribbit_debug_level_counters.argtypes = [C.POINTER(C.c_int64 * 3)]
    L.ribbit_debug_small_motif_counters.restype = None
    L.ribbit_debug_small_motif_counters.argtypes = [C.POINTER(C.c_int64 * 2)]
    L.ribbit_hip_small_motifs.argtypes = [vp, vp, vp, vp, vp, vp]
    L.ribbit_debug_set_merge_min_range.restype = None
    L.ribbit_debug_set_merge_min_range.argtypes = [C.c_size_t]
    L.ribbit_host_longest_runs.argtypes = [C.POINTER(ScanParams), i64, vp, vp, vp, C.c_size_t, vp, C.c_size_t, vp]
    L.ribbit_hip_plane_bits.argtypes = [vp, i32, i64, i64, vp]
    L.ribbit_hip_range_popcount.argtypes = [vp, i32, i64, i64, C.POINTER(i32)]
    L.ribbit_hip_plane_words.restype = i64
    L.ribbit_hip_plane_words.argtypes = [vp]
    L.ribbit_hip_packed_plane.argtypes = [vp, C.c_int, vp]
    L.ribbit_hip_last_timing_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    L.ribbit_refine_params_default.restype = None
    L.ribbit_refine_params_default.argtypes = [C.POINTER(RefineParams), i32, i32]
    L.ribbit_hip_seed_longest_runs.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ribbit_hip_refine_jobs.argtypes = [vp, C.POINTER(RefineParams), C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp)]
    L.ribbit_host_refine_jobs.argtypes = [C.POINTER(ScanParams), C.POINTER(RefineParams), i64, vp, vp, vp, C.c_size_t,
                                          vp, C.c_size_t, vp, C.c_size_t, C.POINTER(vp), C.POINTER(C.c_size_t),
                                          C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ribbit_refine_jobs_free.restype = None
    L.ribbit_refine_jobs_free.argtypes = [vp, vp]
    L.ribbit_hip_refine_bed.argtypes = [vp, C.POINTER(RefineParams), C.c_char_p, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ribbit_host_refine_bed.argtypes = [C.POINTER(ScanParams), C.POINTER(RefineParams), C.c_char_p, i64, vp, vp, vp, C.c_size_t,
                                         vp, C.c_size_t, vp, C.c_size_t, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ribbit_text_free.restype = None
    L.ribbit_text_free.argtypes = [vp]
    L.ribbit_hip_xa_words.argtypes = [vp, i64, i64, vp]
    L.ribbit_hip_perfect_runs_partial.argtypes = [vp, i64, i64, i64, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ribbit_hip_scan_perfect_chunk.argtypes = [vp, i64, i64, i64, vp, C.c_size_t, vp, C.c_size_t,
                                                C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ribbit_hip_host_register.argtypes = [vp, C.c_size_t]
    L.ribbit_hip_host_unregister.argtypes = [vp]
    L.ribbit_hip_set_host_threads.argtypes = [vp, i32]
    L.ribbit_hip_set_timing.argtypes = [vp, i32]
    L.ribbit_hip_debug_set_event_capacity.argtypes = [vp, C.c_size_t]
    L.ribbit_hip_debug_pair_events.argtypes = [vp, vp, C.c_size_t, i64, vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
    L.ribbit_hip_ssw_passes.argtypes = [vp, vp, C.c_size_t, C.c_char_p, C.c_size_t, i32, vp]
    L.ribbit_hip_ssw_align_jobs.argtypes = [vp, vp, C.c_size_t, C.c_char_p, C.c_size_t, i32, vp, vp, C.c_size_t, vp, vp]
    L.ribbit_hip_scan_perfect_begin.argtypes = [vp, i64, i64, i64]
    L.ribbit_hip_scan_perfect_end.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_int,
                                              C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ribbit_hip_scan_perfect_wait.argtypes = [vp]
    L.ribbit_hip_scan_perfect_end_device.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ribbit_host_perfect_runs_from_events.argtypes = [C.POINTER(ScanParams), C.c_size_t, vp, vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ribbit_runs_free.restype = None
    L.ribbit_runs_free.argtypes = [vp]
    L.ribbit_ssw_align.argtypes = [C.c_char_p, i32, C.c_char_p, i32, i32, C.POINTER(Alignment), C.c_char_p, C.c_size_t]
    L.ribbit_debug_ssw_align_periodic.argtypes = [C.c_char_p, i32, C.c_char_p, i32, i32, i32, C.POINTER(Alignment), C.c_char_p, C.c_size_t]
    L.ribbit_hip_stage_calls_chunk.argtypes = [vp, C.c_int, i64, i64, i64, i64, C.POINTER(ChunkCalls)]
    L.ribbit_hip_xa_words_strided.argtypes = [vp, i64, i64, vp, i64]
    L.ribbit_host_merge_chunks.argtypes = [C.POINTER(ScanParams), i64, vp, vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(ChunkPart), C.c_size_t,
                                           C.POINTER(SeedLists)]
    L.ribbit_hip_debug_stream_read.argtypes = [vp, i64, C.POINTER(i64)]
    L.ribbit_hip_last_event_count.restype = i64
    L.ribbit_hip_last_event_count.argtypes = [vp]
    _lib = L
    return L


def _copy(ptr, n, dt):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dt)
    buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt).copy()


def _view(ptr, n, dt):
    """read-only numpy view of n records at ptr (no copy)"""
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dt)
    v = np.frombuffer((C.c_char * (n * dt.itemsize)).from_address(ptr), dtype=dt)
    v.flags.writeable = False
    return v


def pack_planes(sequence: bytes, max_motif: int):
    """numpy packing of a record into (hi, lo, brk) uint32 planes with the padding
    ribbit_host_replay_calls needs.  Host-side helper for replaying call lists that came from
    another rank; scans never use it (the GPU packs its own planes)."""
    n = len(sequence)
    raw = np.frombuffer(sequence, dtype=np.uint8)
    up = raw | 0x20
    valid = (up == ord("a")) | (up == ord("c")) | (up == ord("g")) | (up == ord("t"))
    code = np.where(valid, ((raw >> 1) & 3) ^ ((raw >> 2) & 1), 0).astype(np.uint8)
    nwords = n // 32 + 1 + (max_motif + 2) // 32 + 4
    def pack(bits):
        buf = np.zeros(nwords * 32, dtype=np.uint8)
        buf[:n] = bits
        return np.packbits(buf, bitorder="little").view("<u4").copy()
    brk_bits = np.ones(nwords * 32, dtype=np.uint8)
    brk_bits[:n] = ~valid
    brk = np.packbits(brk_bits, bitorder="little").view("<u4").copy()
    return pack(code >> 1), pack(code & 1), brk


def pack_bit_planes(planes_bits, length: int):
    """list of byte-per-base 0/1 arrays -> (uint32 words motif-major, stride): the xa argument of
    ribbit_host_replay_calls."""
    stride = (length // 32 + 1 + 7) // 8 * 8 + 16
    out = np.zeros((len(planes_bits), stride), dtype="<u4")
    for i, bits in enumerate(planes_bits):
        buf = np.zeros(stride * 32, dtype=np.uint8)
        buf[:length] = bits
        out[i] = np.packbits(buf, bitorder="little").view("<u4")
    return np.ascontiguousarray(out), stride


def host_replay_calls(min_motif: int, max_motif: int, sequence: bytes, perfect_calls, subst_calls=None,
                      anchored_calls=None, xa=None, xa_stride: int = 0):
    """ribbit_host_replay_calls: call lists -> dict(perfect, subst, anchored, dispatch, guard_hits). No GPU needed."""
    L = load_library()
    params = ScanParams()
    L.ribbit_scan_params_default(C.byref(params), min_motif, max_motif)
    hi, lo, brk = pack_planes(sequence, max_motif)
    empty = np.zeros(0, CALL_DT)
    pc = np.ascontiguousarray(perfect_calls, dtype=CALL_DT)
    sc = np.ascontiguousarray(subst_calls if subst_calls is not None else empty, dtype=CALL_DT)
    ac = np.ascontiguousarray(anchored_calls if anchored_calls is not None else empty, dtype=CALL_DT)
    out = SeedLists()
    rc = L.ribbit_host_replay_calls(C.byref(params), len(sequence), hi.ctypes.data, lo.ctypes.data, brk.ctypes.data, len(hi),
                                    xa.ctypes.data if xa is not None else None, xa_stride,
                                    pc.ctypes.data, len(pc), sc.ctypes.data, len(sc),
                                    ac.ctypes.data if anchored_calls is not None else None, len(ac), C.byref(out))
    if rc != 0:
        raise RibbitHipError(f"ribbit_host_replay_calls error {rc}: {L.ribbit_hip_last_error().decode()}")
    try:
        return {"perfect": _copy(out.perfect, out.n_perfect, SEED_DT), "subst": _copy(out.subst, out.n_subst, SEED_DT),
                "anchored": _copy(out.anchored, out.n_anchored, SEED_DT), "dispatch": _copy(out.dispatch, out.n_dispatch, SEED_DT),
                "guard_hits": int(out.guard_hits)}
    finally:
        L.ribbit_seed_lists_free(C.byref(out))


def host_merge_chunks(min_motif: int, max_motif: int, length: int, hi, lo, brk, xa, xa_stride: int, parts):
    """ribbit_host_merge_chunks: the merging rank's half of the chunk-sharded path.  parts (chunk order) = dicts with
    RUN_DT arrays `runs`, `halves` and, per window stage s in ("subst", "anchored"), `<s>_calls` (CALL_DT), `<s>_pend`
    (int32 per call, or None), `<s>_tail_pend` (int), `<s>_flush` (CALL_DT).  -> dict of the record's seed lists."""
    L = load_library()
    params = ScanParams()
    L.ribbit_scan_params_default(C.byref(params), min_motif, max_motif)
    hi, lo, brk = (np.ascontiguousarray(a, dtype="<u4") for a in (hi, lo, brk))
    keep = []                                        # arrays the C structs point into

    def arr(a, dt):
        a = np.ascontiguousarray(a if a is not None else np.zeros(0, dt), dtype=dt)
        keep.append(a)
        return a

    cparts = (ChunkPart * max(len(parts), 1))()
    for k, p in enumerate(parts):
        runs, halves = arr(p["runs"], RUN_DT), arr(p["halves"], RUN_DT)
        cparts[k].runs, cparts[k].n_runs = runs.ctypes.data, len(runs)
        cparts[k].halves, cparts[k].n_halves = halves.ctypes.data, len(halves)
        for stage in ("subst", "anchored"):
            cc = getattr(cparts[k], stage)
            calls, flush = arr(p[f"{stage}_calls"], CALL_DT), arr(p[f"{stage}_flush"], CALL_DT)
            cc.calls, cc.n = calls.ctypes.data, len(calls)
            cc.flush, cc.n_flush = flush.ctypes.data, len(flush)
            pend = p.get(f"{stage}_pend")
            if pend is not None and len(calls):
                pend = arr(pend, "<i4")
                assert len(pend) == len(calls)
                cc.pend = pend.ctypes.data
            cc.tail_pend = int(p[f"{stage}_tail_pend"])
            cc.inexact = int(p.get(f"{stage}_inexact", 0))
    out = SeedLists()
    rc = L.ribbit_host_merge_chunks(C.byref(params), length, hi.ctypes.data, lo.ctypes.data, brk.ctypes.data, len(hi),
                                    xa.ctypes.data if xa is not None else None, xa_stride, cparts, len(parts), C.byref(out))
    if rc != 0:
        raise RibbitHipError(f"ribbit_host_merge_chunks error {rc}: {L.ribbit_hip_last_error().decode()}")
    try:
        return {"perfect": _copy(out.perfect, out.n_perfect, SEED_DT), "subst": _copy(out.subst, out.n_subst, SEED_DT),
                "anchored": _copy(out.anchored, out.n_anchored, SEED_DT), "dispatch": _copy(out.dispatch, out.n_dispatch, SEED_DT),
                "guard_hits": int(out.guard_hits)}
    finally:
        L.ribbit_seed_lists_free(C.byref(out))


def ssw_align(query: bytes, ref: bytes, ref_len: int | None = None, mask_len: int = 15):
    """ribbit_ssw_align -> (dict of alignment fields, cigar string)"""
    L = load_library()
    out = Alignment()
    cap = 16 * (len(query) + len(ref)) + 64
    buf = C.create_string_buffer(cap)
    rc = L.ribbit_ssw_align(query, len(query), ref, len(ref) if ref_len is None else ref_len, mask_len, C.byref(out), buf, cap)
    if rc != 0:
        raise RibbitHipError(f"ribbit_ssw_align error {rc}: {L.ribbit_hip_last_error().decode()}")
    return {n: getattr(out, n) for n, _ in Alignment._fields_}, buf.value.decode()


def ssw_align_periodic(query: bytes, motif: bytes, ref_len: int, mask_len: int = 15):
    """ribbit_debug_ssw_align_periodic (test hook): the alignment against `motif` repeated past ref_len, finished the way
    refinement finishes an alignment whose path is known -> (dict of alignment fields, cigar string)"""
    L = load_library()
    out = Alignment()
    cap = 16 * (len(query) + ref_len + len(motif)) + 64
    buf = C.create_string_buffer(cap)
    rc = L.ribbit_debug_ssw_align_periodic(query, len(query), motif, len(motif), ref_len, mask_len, C.byref(out), buf, cap)
    if rc != 0:
        raise RibbitHipError(f"ribbit_debug_ssw_align_periodic error {rc}: {L.ribbit_hip_last_error().decode()}")
    return {n: getattr(out, n) for n, _ in Alignment._fields_}, buf.value.decode()


def _jobs_with_motifs(jobs, pool: bytes):
    """-> list of (job record, motif string)"""
    return [(j, pool[int(j["motif_offset"]):int(j["motif_offset"]) + int(j["atomicity"])].decode()) for j in jobs]


def alignment_counters():
    """(alignments refinement made, of them with GPU passes, with GPU paths), cumulative"""
    L = load_library()
    out = (C.c_int64 * 3)()
    L.ribbit_debug_alignment_counters(C.byref(out))
    return int(out[0]), int(out[1]), int(out[2])


def level_counters():
    """(levels run, nodes put off, alignments of those nodes) of the level-by-level GPU refinement of long-motif seeds'
    recursion trees, cumulative"""
    L = load_library()
    out = (C.c_int64 * 3)()
    L.ribbit_debug_level_counters(C.byref(out))
    return int(out[0]), int(out[1]), int(out[2])


def last_device_merge():
    """(ranges the GPU merged, ranges it was given but left to the host threads, ranges the host threads merged while its kernel
    ran, ranges of the stage, ranges merged again by the validation walk) of the calling thread's last anchored-stage merge; the
    first three are zero when the merge ran on the host threads alone"""
    L = load_library()
    out = (C.c_int32 * 5)()
    L.ribbit_debug_last_device_merge(C.byref(out))
    return tuple(int(v) for v in out)


def small_motif_counters():
    """(seeds refinement took from the GPU's possibleMotifs table, seeds it computed on the host), cumulative"""
    L = load_library()
    out = (C.c_int64 * 2)()
    L.ribbit_debug_small_motif_counters(C.byref(out))
    return int(out[0]), int(out[1])


def host_longest_runs(min_motif: int, max_motif: int, sequence: bytes, seeds):
    """ribbit_host_longest_runs: longestContinuousMatches of seeds on the composed planes, recomputed on the host. No GPU."""
    L = load_library()
    params = ScanParams()
    L.ribbit_scan_params_default(C.byref(params), min_motif, max_motif)
    hi, lo, brk = pack_planes(sequence, max_motif)
    s = np.ascontiguousarray(seeds, dtype=SEED_DT)
    out = np.zeros(len(s), dtype="<i4")
    rc = L.ribbit_host_longest_runs(C.byref(params), len(sequence), hi.ctypes.data, lo.ctypes.data, brk.ctypes.data, len(hi),
                                    s.ctypes.data, len(s), out.ctypes.data)
    if rc != 0:
        raise RibbitHipError(f"ribbit_host_longest_runs error {rc}: {L.ribbit_hip_last_error().decode()}")
    return out


def host_refine_jobs(min_motif: int, max_motif: int, sequence: bytes, xa, xa_stride: int, dispatch, refine_params=None):
    """ribbit_host_refine_jobs: dispatch seeds + planes -> (jobs array, motif pool bytes). No GPU needed."""
    L = load_library()
    params = ScanParams()
    L.ribbit_scan_params_default(C.byref(params), min_motif, max_motif)
    rp = refine_params
    if rp is None:
        rp = RefineParams()
        L.ribbit_refine_params_default(C.byref(rp), min_motif, max_motif)
    hi, lo, brk = pack_planes(sequence, max_motif)
    d = np.ascontiguousarray(dispatch, dtype=SEED_DT)
    jobs, nj, pool, npool = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
    rc = L.ribbit_host_refine_jobs(C.byref(params), C.byref(rp), len(sequence), hi.ctypes.data, lo.ctypes.data, brk.ctypes.data,
                                   len(hi), xa.ctypes.data if xa is not None else None, xa_stride, d.ctypes.data, len(d),
                                   C.byref(jobs), C.byref(nj), C.byref(pool), C.byref(npool))
    if rc != 0:
        raise RibbitHipError(f"ribbit_host_refine_jobs error {rc}: {L.ribbit_hip_last_error().decode()}")
    try:
        return _copy(jobs.value, nj.value, JOB_DT), C.string_at(pool.value, npool.value)
    finally:
        L.ribbit_refine_jobs_free(jobs, pool)


def host_refine_bed(min_motif: int, max_motif: int, sequence: bytes, xa, xa_stride: int, dispatch, sequence_id: str = "seq",
                    refine_params=None) -> str:
    """ribbit_host_refine_bed: dispatch seeds + planes -> BED text.  No GPU needed."""
    L = load_library()
    params = ScanParams()
    L.ribbit_scan_params_default(C.byref(params), min_motif, max_motif)
    rp = refine_params
    if rp is None:
        rp = RefineParams()
        L.ribbit_refine_params_default(C.byref(rp), min_motif, max_motif)
    hi, lo, brk = pack_planes(sequence, max_motif)
    d = np.ascontiguousarray(dispatch, dtype=SEED_DT)
    text, n = C.c_void_p(), C.c_size_t()
    rc = L.ribbit_host_refine_bed(C.byref(params), C.byref(rp), sequence, len(sequence), hi.ctypes.data, lo.ctypes.data,
                                  brk.ctypes.data, len(hi), xa.ctypes.data if xa is not None else None, xa_stride, d.ctypes.data, len(d),
                                  sequence_id.encode(), C.byref(text), C.byref(n))
    if rc != 0:
        raise RibbitHipError(f"ribbit_host_refine_bed error {rc}: {L.ribbit_hip_last_error().decode()}")
    try:
        return C.string_at(text.value, n.value).decode()
    finally:
        L.ribbit_text_free(text)


def host_perfect_runs_from_events(min_motif: int, max_motif: int, event_parts, count_parts):
    """ribbit_host_perfect_runs_from_events: per-rank (events, per-motif counts) -> paired runs. No GPU needed."""
    L = load_library()
    params = ScanParams()
    L.ribbit_scan_params_default(C.byref(params), min_motif, max_motif)
    ev = np.ascontiguousarray(np.concatenate(event_parts) if len(event_parts) else np.zeros(0, "<u8"), dtype="<u8")
    cnt = np.ascontiguousarray(np.concatenate(count_parts), dtype="<u8")
    runs, n = C.c_void_p(), C.c_size_t()
    rc = L.ribbit_host_perfect_runs_from_events(C.byref(params), len(event_parts), ev.ctypes.data, cnt.ctypes.data, C.byref(runs), C.byref(n))
    if rc != 0:
        raise RibbitHipError(f"ribbit_host_perfect_runs_from_events error {rc}: {L.ribbit_hip_last_error().decode()}")
    try:
        return _copy(runs.value, n.value, RUN_DT)
    finally:
        L.ribbit_runs_free(runs)


def pair_halves(halves) -> np.ndarray:
    """Cross-rank pairing of the edge events of chunk-local pairing (Scanner.perfect_runs_partial): sorted by
    (motif, position) they alternate START, END."""
    hv = np.sort(np.asarray(halves, dtype="<u8").view("<u8"), kind="stable") if len(halves) else np.zeros(0, "<u8")
    if len(hv) == 0:
        return np.zeros(0, RUN_DT)
    pos = (hv & np.uint64(0xFFFFFFFF)).astype(np.int64)
    mlen = ((hv >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
    kind = ((hv >> np.uint64(48)) & np.uint64(0xF)).astype(np.int64)
    order = np.lexsort((pos, mlen))
    pos, mlen, kind = pos[order], mlen[order], kind[order]
    if len(hv) % 2 or np.any(kind[0::2] != 0) or np.any(kind[1::2] == 0) or np.any(mlen[0::2] != mlen[1::2]):
        raise RibbitHipError("edge events of the chunks do not pair up")
    out = np.zeros(len(hv) // 2, RUN_DT)
    out["start"], out["end"], out["mlen"] = pos[0::2], pos[1::2], mlen[0::2]
    out["term"] = kind[1::2] - 1
    return out


RUN_NOT_OWNED, RUN_HALF_START, RUN_HALF_END = -1, 3, 4


def join_run_halves(halves) -> np.ndarray:
    """Half records of all chunks of one record (Scanner.scan_perfect_chunk) -> the runs they form: ordered by
    (motif, position) every open START is closed by the next orphan END of its motif."""
    hv = np.concatenate([np.asarray(h, dtype=RUN_DT) for h in halves]) if len(halves) else np.zeros(0, RUN_DT)
    hs = hv[hv["term"] == RUN_HALF_START]
    he = hv[hv["term"] >= RUN_HALF_END]
    if len(hs) + len(he) != len(hv) or len(hs) != len(he):
        raise RibbitHipError(f"{len(hs)} open run starts but {len(he)} orphan run ends across the chunks")
    hs = hs[np.lexsort((hs["start"], hs["mlen"]))]
    he = he[np.lexsort((he["end"], he["mlen"]))]
    if np.any(hs["mlen"] != he["mlen"]) or np.any(he["end"] <= hs["start"]) or \
            np.any((hs["mlen"][1:] == hs["mlen"][:-1]) & (hs["start"][1:] <= he["end"][:-1])):
        raise RibbitHipError("run halves of the chunks do not pair up")
    joined = np.zeros(len(hs), RUN_DT)
    joined["start"], joined["end"], joined["mlen"], joined["term"] = hs["start"], he["end"], hs["mlen"], he["term"] - RUN_HALF_END
    return joined


def merge_chunk_runs(parts, halves) -> np.ndarray:
    """All chunks' run records and halves -> the record's runs ordered by (mlen, start) (copies; for tests and
    small inputs -- a streaming consumer walks the parts in place and skips term == RUN_NOT_OWNED)."""
    allr = np.concatenate([np.asarray(p) for p in parts] + [join_run_halves(halves)])
    allr = allr[allr["term"] >= 0]
    return allr[np.lexsort((allr["start"], allr["mlen"]))]


def read_fasta(path: str, pinned: bool = False):
    """records of a FASTA file as ribbit_fasta_* delimits them (the reference's reader loop, ribbit.cpp:269-280):
    [(name, bases bytes, is_last)]"""
    L = load_library()
    r = C.c_void_p()
    if L.ribbit_fasta_open(path.encode(), int(pinned), C.byref(r)) != 0:
        raise RibbitHipError(L.ribbit_fasta_last_error().decode())
    out = []
    try:
        while True:
            name, bases, n, last = C.c_char_p(), C.c_void_p(), C.c_int64(), C.c_int()
            got = L.ribbit_fasta_next(r, C.byref(name), C.byref(bases), C.byref(n), C.byref(last))
            if got < 0:
                raise RibbitHipError(L.ribbit_fasta_last_error().decode())
            if got == 0:
                break
            out.append((name.value.decode(), C.string_at(bases.value, n.value), bool(last.value)))
            L.ribbit_fasta_release(r, bases)
    finally:
        L.ribbit_fasta_close(r)
    return out


class PinnedBuffer:
    """page-locked host memory (ribbit_hip_host_alloc) as a writable numpy uint8 view"""

    def __init__(self, nbytes: int):
        self._L = load_library()
        p = C.c_void_p()
        rc = self._L.ribbit_hip_host_alloc(max(int(nbytes), 1), C.byref(p))
        if rc != 0:
            raise RibbitHipError(f"ribbit_hip_host_alloc error {rc}: {self._L.ribbit_hip_last_error().decode()}")
        self.ptr = p.value
        self.nbytes = int(nbytes)
        self.array = np.frombuffer((C.c_char * max(self.nbytes, 1)).from_address(self.ptr), dtype=np.uint8)[:self.nbytes]

    def close(self):
        if self.ptr:
            self.array = None
            self._L.ribbit_hip_host_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Scanner:
    """One GPU-resident FASTA record and the scans over it.

    Method names follow the reference functions they stand in for
    (parse_perfect_shiftxor.h:10, fasta_utils.cpp:59-250).
    """

    def __init__(self, min_motif: int = 2, max_motif: int = 100, device: int = 0):
        self._L = load_library()
        self.params = ScanParams()
        self._L.ribbit_scan_params_default(C.byref(self.params), min_motif, max_motif)
        h = C.c_void_p()
        self._h = None
        self._check(self._L.ribbit_hip_open(C.byref(self.params), device, C.byref(h)))
        self._h = h
        self.length = 0
        self.min_shift = min_motif - 2 if min_motif > 2 else 1     # ribbit.cpp:241
        self.max_shift = max_motif + 2                              # ribbit.cpp:242

    def _check(self, rc):
        if rc != 0:
            raise RibbitHipError(f"ribbit_hip error {rc}: {self._L.ribbit_hip_last_error().decode()}")

    def close(self):
        if self._h is not None:
            self._L.ribbit_hip_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def debug_pair_events(self, events: np.ndarray, length: int):
        """-> (runs, flags): the device-side pairing on a caller-made event stream (uint64 events in buffer order)"""
        ev = np.ascontiguousarray(events, dtype="<u8")
        runs = np.zeros(max(len(ev), 1), RUN_DT)
        n, flags = C.c_size_t(), C.c_uint32()
        self._check(self._L.ribbit_hip_debug_pair_events(self._h, ev.ctypes.data, len(ev), length, runs.ctypes.data, len(runs), C.byref(n), C.byref(flags)))
        return runs[:min(n.value, len(runs))], flags.value

    def debug_set_event_capacity(self, events: int) -> None:
        self._check(self._L.ribbit_hip_debug_set_event_capacity(self._h, events))

    def set_timing(self, enabled: bool) -> None:
        self._check(self._L.ribbit_hip_set_timing(self._h, int(enabled)))

    def set_stream(self, hip_stream: int | None):
        self._check(self._L.ribbit_hip_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    # fasta_utils.cpp:78-122 -------------------------------------------------------------
    def load_record(self, sequence: bytes):
        self._keep = bytes(sequence)
        self.length = len(self._keep)
        self._check(self._L.ribbit_hip_load_record(self._h, self._keep, self.length))

    def load_record_pinned(self, host_ptr: int, length: int):
        """bases in page-locked memory (PinnedBuffer) that stays valid until the next load: async upload, no host copy"""
        self.length = int(length)
        self._check(self._L.ribbit_hip_load_record_pinned(self._h, C.c_void_p(host_ptr), self.length))

    def load_record_device(self, dev_ptr: int, length: int):
        self.length = int(length)
        self._check(self._L.ribbit_hip_load_record_device(self._h, C.c_void_p(dev_ptr), self.length))

    def _list(self, fn, dt, copy=True):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(fn(self._h, C.byref(p), C.byref(n)))
        return _copy(p.value, n.value, dt) if copy else _view(p.value, n.value, dt)

    # parse_perfect_shiftxor.cpp:146-226 ---------------------------------------------------
    def scan_perfect_runs(self, copy=True):
        """Runs of the perfect scan ordered by (mlen, start).  copy=False returns a read-only view of the
        library's pinned result buffer, valid until the next call on this Scanner (what a C caller gets)."""
        return self._list(self._L.ribbit_hip_scan_perfect_runs, RUN_DT, copy)

    def perfect_calls(self):
        return self._list(self._L.ribbit_hip_perfect_calls, CALL_DT)

    def processShiftXORsPerfect(self, copy=True):
        """copy=False: a read-only view of the library's list, valid until the next call on this Scanner (what a C caller gets)"""
        return self._list(self._L.ribbit_hip_seeds_perfect, SEED_DT, copy)

    # parse_substitute_shiftxor.cpp:391-577 ------------------------------------------------
    def subst_calls(self):
        return self._list(self._L.ribbit_hip_subst_calls, CALL_DT)

    def processShiftXORswithSubstitutions(self):
        """-> (seed_positions_perfect as re-typed by this stage, seed_positions_substut)"""
        pp, np_, ps, ns = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
        self._check(self._L.ribbit_hip_seeds_substitutions(self._h, C.byref(pp), C.byref(np_), C.byref(ps), C.byref(ns)))
        return _copy(pp.value, np_.value, SEED_DT), _copy(ps.value, ns.value, SEED_DT)

    # parse_anchored_shiftxor.cpp:20-56, 538-726; fasta_utils.cpp:143-224 --------------------
    def anchored_calls(self):
        return self._list(self._L.ribbit_hip_anchored_calls, CALL_DT)

    def processShiftXORsAnchored(self, copy=True):
        """-> (perfect, substitution, anchored seed lists as the anchored stage leaves them); copy=False: read-only views of
        the library's lists, valid until the next call on this Scanner (what a C caller gets: three pointers)"""
        ptrs = [C.c_void_p() for _ in range(3)]
        ns = [C.c_size_t() for _ in range(3)]
        args = []
        for p, n in zip(ptrs, ns):
            args += [C.byref(p), C.byref(n)]
        self._check(self._L.ribbit_hip_seeds_anchored(self._h, *args))
        if copy:
            return tuple(_copy(p.value, n.value, SEED_DT) for p, n in zip(ptrs, ns))
        return tuple(_view(p.value, n.value, SEED_DT) for p, n in zip(ptrs, ns))

    def dispatch_seeds(self, copy=True):
        return self._list(self._L.ribbit_hip_dispatch_seeds, SEED_DT, copy)

    # parse_seed.cpp:26-44 (batched on the GPU), parse_smallmotif_seed.cpp:76-270, parse_seed.cpp:153-404 ----
    def seed_longest_runs(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._L.ribbit_hip_seed_longest_runs(self._h, C.byref(p), C.byref(n)))
        return _copy(p.value, n.value, np.dtype("<i4"))

    def refine_jobs(self, refine_params=None):
        rp = refine_params
        if rp is None:
            rp = RefineParams()
            self._L.ribbit_refine_params_default(C.byref(rp), self.params.min_motif, self.params.max_motif)
        jobs, n, pool = C.c_void_p(), C.c_size_t(), C.c_void_p()
        self._check(self._L.ribbit_hip_refine_jobs(self._h, C.byref(rp), C.byref(jobs), C.byref(n), C.byref(pool)))
        arr = _copy(jobs.value, n.value, JOB_DT)
        size = int((arr["motif_offset"] + arr["atomicity"]).max()) if len(arr) else 0
        return arr, (C.string_at(pool.value, size) if size else b"")

    def small_motifs(self, refine_params=None):
        """ribbit_hip_small_motifs: (head [n_seeds, 4] int32, records [n_records, 4] uint32) -- possibleMotifs of the
        dispatched seeds with m <= 10 as the GPU computed it"""
        rp = refine_params
        if rp is None:
            rp = RefineParams()
            self._L.ribbit_refine_params_default(C.byref(rp), self.params.min_motif, self.params.max_motif)
        head, n, rec, nr = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
        self._check(self._L.ribbit_hip_small_motifs(self._h, C.byref(rp), C.byref(head), C.byref(n), C.byref(rec), C.byref(nr)))
        return _copy(head.value, 4 * n.value, np.dtype("<i4")).reshape(-1, 4), _copy(rec.value, 4 * nr.value, np.dtype("<u4")).reshape(-1, 4)

    def refine_bed(self, sequence_id: str = "seq", refine_params=None) -> str:
        """BED rows of the loaded record (fasta_utils.cpp:211-242 and everything below it)"""
        rp = refine_params
        if rp is None:
            rp = RefineParams()
            self._L.ribbit_refine_params_default(C.byref(rp), self.params.min_motif, self.params.max_motif)
        text, n = C.c_void_p(), C.c_size_t()
        self._check(self._L.ribbit_hip_refine_bed(self._h, C.byref(rp), sequence_id.encode(), C.byref(text), C.byref(n)))
        return C.string_at(text.value, n.value).decode()

    def refine_bed_view(self, sequence_id: str = "seq", refine_params=None) -> np.ndarray:
        """The same BED text as the C ABI hands it out: a uint8 view of the library's buffer (valid until the handle's next
        refine_bed call or its close), without the copy and the decoding into a Python string that refine_bed adds -- 150 MB
        twice over for a chromosome, which a C caller (ribbit-hip writes the buffer straight to its output file) never pays."""
        rp = refine_params
        if rp is None:
            rp = RefineParams()
            self._L.ribbit_refine_params_default(C.byref(rp), self.params.min_motif, self.params.max_motif)
        text, n = C.c_void_p(), C.c_size_t()
        self._check(self._L.ribbit_hip_refine_bed(self._h, C.byref(rp), sequence_id.encode(), C.byref(text), C.byref(n)))
        if not n.value:
            return np.zeros(0, dtype=np.uint8)
        return np.ctypeslib.as_array(C.cast(text.value, C.POINTER(C.c_uint8)), shape=(n.value,))

    def adopt_dispatch(self, seeds) -> None:
        """ribbit_hip_adopt_dispatch: this handle (same record loaded) refines a slice of another handle's dispatch list"""
        d = np.ascontiguousarray(seeds, dtype=SEED_DT)
        self._check(self._L.ribbit_hip_adopt_dispatch(self._h, d.ctypes.data, len(d)))

    def refine_met_empty_query(self) -> bool:
        return bool(self._L.ribbit_hip_refine_met_empty_query(self._h))

    def guard_hits(self) -> int:
        return int(self._L.ribbit_hip_guard_hits(self._h))

    # chunk-sharded operation --------------------------------------------------------------
    def perfect_runs_partial(self, own_lo: int, own_hi: int, pos_offset: int = 0):
        """-> (complete runs of this chunk, unmatched edge events as uint64)"""
        r, nr, hv, nh = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
        self._check(self._L.ribbit_hip_perfect_runs_partial(self._h, own_lo, own_hi, pos_offset, C.byref(r), C.byref(nr), C.byref(hv), C.byref(nh)))
        return _copy(r.value, nr.value, RUN_DT), _copy(hv.value, nh.value, np.dtype("<u8"))

    def scan_perfect_chunk(self, own_lo: int, own_hi: int, pos_offset: int = 0, out: np.ndarray | None = None,
                           halves_out: np.ndarray | None = None):
        """ribbit_hip_scan_perfect_chunk: device-paired run records of one chunk.
        Without out -> (read-only view of the library's pinned run records, copy of the halves).
        With out / halves_out (RUN_DT arrays to fill in place, e.g. over registered shared memory) -> (n, n_halves).
        Records with term == RUN_NOT_OWNED are place holders to skip."""
        p, n, hp, nh = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
        if out is not None:
            assert out.dtype == RUN_DT and out.flags.c_contiguous and halves_out is not None and halves_out.dtype == RUN_DT
            self._check(self._L.ribbit_hip_scan_perfect_chunk(self._h, own_lo, own_hi, pos_offset, out.ctypes.data, len(out),
                                                              halves_out.ctypes.data, len(halves_out),
                                                              C.byref(p), C.byref(n), C.byref(hp), C.byref(nh)))
            return n.value, nh.value
        self._check(self._L.ribbit_hip_scan_perfect_chunk(self._h, own_lo, own_hi, pos_offset, None, 0, None, 0,
                                                          C.byref(p), C.byref(n), C.byref(hp), C.byref(nh)))
        halves = _copy(hp.value, nh.value, RUN_DT)
        if n.value == 0:
            return np.zeros(0, RUN_DT), halves
        view = np.frombuffer((C.c_char * (n.value * RUN_DT.itemsize)).from_address(p.value), dtype=RUN_DT)
        view.flags.writeable = False
        return view, halves

    def scan_perfect_begin(self, own_lo: int = 0, own_hi: int = (1 << 63) - 1, pos_offset: int = 0) -> None:
        """Enqueue the perfect scan of the loaded record (or of one chunk of it) and return without waiting."""
        self._check(self._L.ribbit_hip_scan_perfect_begin(self._h, own_lo, own_hi, pos_offset))

    def scan_perfect_end_device(self):
        """Finish the scan begun by scan_perfect_begin WITHOUT copying anything to the host:
        -> (device pointer of the run records, their number, device pointer of the half records, their number).
        The memory is the handle's and stays valid until its next scan (ribbit_hip_scan_perfect_end_device)."""
        p, n, hp, nh = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
        self._check(self._L.ribbit_hip_scan_perfect_end_device(self._h, C.byref(p), C.byref(n), C.byref(hp), C.byref(nh)))
        return p.value or 0, n.value, hp.value or 0, nh.value

    def scan_perfect_wait(self) -> None:
        self._check(self._L.ribbit_hip_scan_perfect_wait(self._h))

    def scan_perfect_end(self, out: np.ndarray | None = None, halves_out: np.ndarray | None = None, wait: bool = True):
        """Finish the scan begun by scan_perfect_begin; returns like scan_perfect_chunk.  wait=False: the records
        (and halves) are still in flight until scan_perfect_wait(); halves are then returned as a view, not a copy."""
        p, n, hp, nh = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
        if out is not None:
            assert out.dtype == RUN_DT and out.flags.c_contiguous and halves_out is not None and halves_out.dtype == RUN_DT
            self._check(self._L.ribbit_hip_scan_perfect_end(self._h, out.ctypes.data, len(out), halves_out.ctypes.data, len(halves_out),
                                                            int(wait), C.byref(p), C.byref(n), C.byref(hp), C.byref(nh)))
            return n.value, nh.value
        self._check(self._L.ribbit_hip_scan_perfect_end(self._h, None, 0, None, 0, int(wait), C.byref(p), C.byref(n), C.byref(hp), C.byref(nh)))
        if wait or nh.value == 0:
            halves = _copy(hp.value, nh.value, RUN_DT)
        else:
            halves = np.frombuffer((C.c_char * (nh.value * RUN_DT.itemsize)).from_address(hp.value), dtype=RUN_DT)
        if n.value == 0:
            return np.zeros(0, RUN_DT), halves
        view = np.frombuffer((C.c_char * (n.value * RUN_DT.itemsize)).from_address(p.value), dtype=RUN_DT)
        view.flags.writeable = False
        return view, halves

    def ssw_passes(self, jobs: np.ndarray, motif_pool: bytes, mask_len: int = 15) -> np.ndarray:
        """ribbit_hip_ssw_passes: the striped passes of every job (JOB_DT) on the GPU -> ENDS_DT records."""
        jobs = np.ascontiguousarray(jobs, dtype=JOB_DT)
        out = np.zeros(len(jobs), ENDS_DT)
        self._check(self._L.ribbit_hip_ssw_passes(self._h, jobs.ctypes.data, len(jobs), motif_pool, len(motif_pool), mask_len, out.ctypes.data))
        return out

    def ssw_align_jobs(self, jobs: np.ndarray, motif_pool: bytes, mask_len: int = 15):
        """ribbit_hip_ssw_align_jobs: whole alignments, passes and path search on the GPU where the kernels take them.
        -> (list of (result dict, cigar string), on_gpu array: 0 host, 1 passes on the GPU, 2 passes and path)"""
        jobs = np.ascontiguousarray(jobs, dtype=JOB_DT)
        n = len(jobs)
        out = (Alignment * max(n, 1))()
        cap = int(16 * (jobs["query_length"].astype(np.int64) + jobs["ppr_length"]).sum()) + 64 * n + 64
        buf = C.create_string_buffer(cap)
        off = np.zeros(max(n, 1), dtype=np.int64)
        on_gpu = np.zeros(max(n, 1), dtype=np.int32)
        self._check(self._L.ribbit_hip_ssw_align_jobs(self._h, jobs.ctypes.data, n, motif_pool, len(motif_pool), mask_len, out, buf, cap,
                                                      off.ctypes.data, on_gpu.ctypes.data))
        res = []
        for j in range(n):
            res.append(({name: getattr(out[j], name) for name, _ in Alignment._fields_}, C.string_at(C.addressof(buf) + int(off[j])).decode()))
        return res, on_gpu[:n]

    def host_register(self, address: int, nbytes: int) -> None:
        """Page-lock caller-owned host memory so that scan_perfect_chunk(out=...) DMAs straight into it."""
        self._check(self._L.ribbit_hip_host_register(address, nbytes))

    def host_unregister(self, address: int) -> None:
        self._check(self._L.ribbit_hip_host_unregister(address))

    def stage_calls_chunk(self, stage: int, own_lo: int, own_hi: int, pos_offset: int, record_length: int) -> dict:
        """ribbit_hip_stage_calls_chunk: the kept calls of one window stage (STAGE_SUBST / STAGE_ANCHORED) of the loaded
        piece, in record coordinates -> dict(calls, pend (or None), tail_pend, flush, inexact, streaks) (copies)"""
        cc = ChunkCalls()
        self._check(self._L.ribbit_hip_stage_calls_chunk(self._h, stage, own_lo, min(own_hi, (1 << 62)), pos_offset, record_length, C.byref(cc)))
        return {"calls": _copy(cc.calls, cc.n, CALL_DT), "pend": _copy(cc.pend, cc.n, np.dtype("<i4")) if cc.pend else None,
                "tail_pend": int(cc.tail_pend), "flush": _copy(cc.flush, cc.n_flush, CALL_DT), "inexact": bool(cc.inexact),
                "streaks": int(cc.streaks)}

    def xa_words_into(self, word_lo: int, word_hi: int, out: np.ndarray, out_word: int) -> None:
        """ribbit_hip_xa_words_strided: this piece's words [word_lo, word_hi) of every composed plane into out[:, out_word:...]
        (out: (motifs, stride) uint32, C-contiguous -- e.g. the record's planes in a segment every rank maps)"""
        assert out.dtype == np.dtype("<u4") and out.flags.c_contiguous and out.ndim == 2
        assert out_word >= 0 and out_word + (word_hi - word_lo) <= out.shape[1]
        self._check(self._L.ribbit_hip_xa_words_strided(self._h, word_lo, word_hi, out.ctypes.data + 4 * out_word, out.shape[1]))

    def xa_words(self, word_lo: int, word_hi: int):
        nm = self.params.max_motif - self.params.min_motif + 1
        out = np.empty((nm, max(word_hi - word_lo, 0)), dtype="<u4")
        self._check(self._L.ribbit_hip_xa_words(self._h, word_lo, word_hi, out.ctypes.data_as(C.c_void_p)))
        return out

    # plane access -----------------------------------------------------------------------
    def plane_bits(self, shift: int, start: int = 0, end: int | None = None):
        end = self.length if end is None else end
        out = np.empty(max(end - start, 0), dtype=np.uint8)
        self._check(self._L.ribbit_hip_plane_bits(self._h, shift, start, end, out.ctypes.data_as(C.c_void_p)))
        return out

    def range_popcount(self, shift: int, start: int, end: int) -> int:
        c = C.c_int32()
        self._check(self._L.ribbit_hip_range_popcount(self._h, shift, start, end, C.byref(c)))
        return c.value

    def packed_plane(self, which: int):
        n = self._L.ribbit_hip_plane_words(self._h)
        out = np.empty(n, dtype=np.uint32)
        self._check(self._L.ribbit_hip_packed_plane(self._h, which, out.ctypes.data_as(C.c_void_p)))
        return out

    def timing_ms(self, what: int) -> float:
        ms = C.c_double()
        self._check(self._L.ribbit_hip_last_timing_ms(self._h, what, C.byref(ms)))
        return ms.value

    def debug_stream_read(self, nbytes: int) -> int:
        n = C.c_int64()
        self._check(self._L.ribbit_hip_debug_stream_read(self._h, nbytes, C.byref(n)))
        return n.value

    def last_event_count(self) -> int:
        return int(self._L.ribbit_hip_last_event_count(self._h))
