"""Chunk-sharded scan of ONE long record over several GPUs (BASELINE.json north_star; SURVEY.md 8e option 2).

Every rank loads its chunk plus halos as a record of its own (a "piece") and runs the WHOLE device side of the three
stages on it: the scan kernels, the pairing of run / pass-streak events, the per-motif window state machines, the stages'
length filters and the call order (window_stage.hip).  A rank keeps the addSeed calls whose scan position it owns, so what
travels to the rank that runs the host merges is

    16 bytes per perfect run record, 16 bytes per KEPT call of the two window stages (nine anchored calls in ten fail
    the length filter and never leave their GPU), a 4-byte cursor bound for the few calls made at an N, and

    the rank's own words of the packed planes (0.375 B/base) and of the composed planes XA_m (max_motif/8 B/base), which
    the merges' range reads need (retainNestedSeed & co, parse_perfect_shiftxor.cpp:18-43,
    parse_anchored_shiftxor.cpp:59-84) -- on one node they go down each GPU's own PCIe link into the merging rank's
    planes, not through a collective.

No event and no streak record is exchanged, and nothing is replayed on the host: the merging rank concatenates the chunks'
lists (they are in call order: chunks own increasing ranges of scan positions) and runs the three order-dependent merges
once, exactly as for a single GPU (ribbit_host_merge_chunks).

Correctness does not depend on the partition.  Scan events are local functions of the sequence, but a CALL is a function
of its whole group of pass-streaks and of the first evaluated window behind it, which a long repeat or a block of N can
stretch beyond any fixed halo.  Every call of ribbit_hip_stage_calls_chunk therefore checks that no call it keeps reaches
the artificial left end of its piece and reports `inexact` otherwise; scan_part then loads the chunk again with a four
times longer left halo (at worst from the start of the record, which is exact by construction)."""
from __future__ import annotations

import numpy as np

import ribbit_amd
from ribbit_amd import CALL_DT, RUN_DT, STAGE_ANCHORED, STAGE_SUBST

_ARRAYS = [("runs", RUN_DT), ("halves", RUN_DT),
           ("subst_calls", CALL_DT), ("subst_pend", np.dtype("<i4")), ("subst_flush", CALL_DT),
           ("anchored_calls", CALL_DT), ("anchored_pend", np.dtype("<i4")), ("anchored_flush", CALL_DT),
           ("hi", np.dtype("<u4")), ("lo", np.dtype("<u4")), ("brk", np.dtype("<u4")), ("xa", np.dtype("<u4"))]
_SCALARS = ["own_lo", "own_hi", "subst_tail_pend", "anchored_tail_pend", "halo_grown", "left_halo"]
CALL_KEYS = ("runs", "halves", "subst_calls", "subst_pend", "subst_flush", "anchored_calls", "anchored_pend", "anchored_flush")
PLANE_KEYS = ("hi", "lo", "brk", "xa")


def plan_chunks(length: int, nparts: int, max_motif: int, left_halo: int | None = None):
    """[(own_lo, own_hi, load_lo, load_hi)] in record coordinates; own ranges are 32-aligned (the planes are exchanged as
    whole words) and the last one includes position `length` (end-of-record calls).  left_halo: first guess of the left
    halo in bases (default 2 (max_motif + 2) + 64 + 4096: a few hundred bases cover the scan kernels' reach, the rest is
    room for the groups of pass-streaks that straddle the cut)."""
    s = max_motif + 2
    halo_left = (2 * s + 64 + 4096) if left_halo is None else max(left_halo, 2 * s + 64)
    halo_right = 4 * s + 64
    cuts = [(length * k // nparts) // 32 * 32 for k in range(nparts)] + [length + 1]
    plans = []
    for k in range(nparts):
        own_lo, own_hi = cuts[k], cuts[k + 1]
        load_lo = max(0, (own_lo - halo_left) // 32 * 32)
        load_hi = min(length, own_hi + halo_right)
        plans.append((own_lo, own_hi, load_lo, load_hi))
    return plans


def scan_part(scanner: "ribbit_amd.Scanner", sequence, plan, length: int | None = None) -> dict:
    """What one rank contributes: perfect runs, kept calls of both window stages, and its own words of all planes.
    sequence: the record's bytes, or a callable (lo, hi) -> bytes for ranks that fetch their pieces."""
    own_lo, own_hi, load_lo, load_hi = plan
    fetch = sequence if callable(sequence) else (lambda lo, hi: sequence[lo:hi])
    length = len(sequence) if length is None else length
    grown = 0
    while True:
        scanner.load_record(fetch(load_lo, load_hi))
        lo_l, hi_l = own_lo - load_lo, own_hi - load_lo
        runs, halves = scanner.scan_perfect_chunk(lo_l, hi_l, load_lo)
        runs = np.array(runs[runs["term"] >= 0])                      # place holders stay behind
        subst = scanner.stage_calls_chunk(STAGE_SUBST, lo_l, hi_l, load_lo, length)
        anchored = scanner.stage_calls_chunk(STAGE_ANCHORED, lo_l, hi_l, load_lo, length)
        if not (subst["inexact"] or anchored["inexact"]):
            break
        # a repeat or a block of N reaches further left than the halo: four times the halo, at worst the whole prefix
        assert load_lo > 0, "a piece that starts where the record starts is exact by construction"
        load_lo = max(0, (own_lo - 4 * (own_lo - load_lo)) // 32 * 32)
        grown += 1
    part = {"own_lo": own_lo, "own_hi": own_hi, "halo_grown": grown, "left_halo": own_lo - load_lo, "runs": runs, "halves": halves}
    for name, st in (("subst", subst), ("anchored", anchored)):
        part[f"{name}_calls"], part[f"{name}_pend"] = st["calls"], st["pend"]
        part[f"{name}_tail_pend"], part[f"{name}_flush"] = st["tail_pend"], st["flush"]
    w_lo = (own_lo - load_lo) // 32
    w_hi = min((own_hi - load_lo + 31) // 32, (load_hi - load_lo) // 32 + 1)
    for which, key in ((0, "hi"), (1, "lo"), (2, "brk")):
        part[key] = scanner.packed_plane(which)[w_lo:w_hi].copy()
    part["xa"] = scanner.xa_words(w_lo, w_hi)
    return part


def part_bytes(part: dict) -> dict:
    """bytes one rank sends: the records of the three stages (what north_star calls the candidate seed intervals) and,
    separately, its words of the planes"""
    size = lambda a: 0 if a is None else int(np.asarray(a).size) * int(np.asarray(a).itemsize)
    kept = len(part["subst_calls"]) + len(part["anchored_calls"])
    return {"records": sum(size(part[k]) for k in CALL_KEYS), "planes": sum(size(part[k]) for k in PLANE_KEYS),
            "kept_window_calls": kept, "perfect_runs": len(part["runs"]),
            "window_call_bytes": size(part["subst_calls"]) + size(part["anchored_calls"])}


def merge_parts(min_motif: int, max_motif: int, length: int, parts: list) -> dict:
    """The merging rank after the exchange: the record's planes from the chunks' words, then the three merges and the
    dispatch merge over the chunks' kept calls (ribbit_host_merge_chunks).  -> seed lists + the planes (for refinement)"""
    nwords = length // 32 + 1 + (max_motif + 2) // 32 + 4
    stride = (length // 32 + 1 + 7) // 8 * 8 + 16
    hi = np.zeros(nwords, "<u4"); lo = np.zeros(nwords, "<u4"); brk = np.full(nwords, 0xFFFFFFFF, "<u4")
    nm = max_motif - min_motif + 1
    xa = np.zeros((nm, stride), "<u4")
    for p in parts:
        w0 = p["own_lo"] // 32
        n = len(p["hi"])
        hi[w0:w0 + n], lo[w0:w0 + n], brk[w0:w0 + n] = p["hi"], p["lo"], p["brk"]
        xa[:, w0:w0 + n] = np.asarray(p["xa"]).reshape(nm, -1)
    lists = ribbit_amd.host_merge_chunks(min_motif, max_motif, length, hi, lo, brk, xa, stride, parts)
    lists["planes"] = (hi, lo, brk, xa, stride)
    return lists


def gather_parts(part: dict, device=None, dst: int = 0):
    """gather-v of every array of `part` to rank `dst` (torch.distributed must be initialised; backend nccl = RCCL over
    xGMI on GPUs, gloo in the CPU tests): the list of all ranks' parts on dst, None elsewhere.  Only dst receives."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")

    def nbytes(a):
        return -1 if a is None else int(np.asarray(a).size) * int(np.asarray(a).itemsize)

    meta = torch.tensor([int(part[k]) for k in _SCALARS] + [nbytes(part.get(k)) for k, _ in _ARRAYS], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    metas = [m.cpu().numpy() for m in metas]
    out = [{k: int(m[j]) for j, k in enumerate(_SCALARS)} for m in metas]
    for j, (k, dt) in enumerate(_ARRAYS):
        sizes = [int(m[len(_SCALARS) + j]) for m in metas]
        cap = max(max(sizes), 1)
        buf = torch.zeros(cap, dtype=torch.uint8, device=dev)
        mine = part.get(k)
        if mine is not None and np.asarray(mine).size:
            raw = np.ascontiguousarray(mine).view(np.uint8).reshape(-1)
            buf[:raw.size] = torch.from_numpy(raw.copy()).to(dev)
        gathered = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
        dist.gather(buf, gathered, dst=dst)
        if rank != dst:
            continue
        for r in range(world):
            out[r][k] = None if sizes[r] < 0 else gathered[r][:sizes[r]].cpu().numpy().view(dt).copy()
    if rank != dst:
        return None
    return out


def gather_parts_shm(part: dict, dst: int = 0):
    """The same gather through host memory shared by the ranks of ONE node (ribbit_amd.node_gather) instead of the collective
    backend: every rank writes its part -- a header of the scalars and the arrays' sizes, then the arrays back to back -- into
    its own cell of a segment all ranks map, and rank `dst` (which must be 0, the segment's consumer) reads them there.  What a
    chunk keeps is host data when it gets here (scan_part copied it off the GPU), so this is N memcpys side by side where
    gather_parts goes host -> device -> gather -> host.  Collective.  -> (ok, parts): ok is the same on every rank (False: the
    segment could not be made or attached, e.g. /dev/shm too small -- use gather_parts); parts as gather_parts returns them."""
    import torch.distributed as dist

    from .distributed import open_node_gather
    assert dst == 0
    world, rank = dist.get_world_size(), dist.get_rank()
    raws = []
    for k, _ in _ARRAYS:
        a = part.get(k)
        raws.append(None if a is None else np.ascontiguousarray(a).view(np.uint8).reshape(-1))
    header = np.array([int(part[k]) for k in _SCALARS] + [-1 if r is None else int(r.size) for r in raws], dtype=np.int64)
    blob = np.concatenate([header.view(np.uint8)] + [r for r in raws if r is not None and r.size])
    sizes = [None] * world
    dist.all_gather_object(sizes, int(blob.size))
    ng = open_node_gather(np.dtype(np.uint8), max(sizes) + 64, 1, nslots=1)
    if ng is None:
        return False, None
    out = None
    try:
        ng.wait_free(1)
        cell, _ = ng.mine(1)
        cell[:blob.size] = blob
        ng.publish(1, int(blob.size), 0)
        if rank == dst:
            cells, _ = ng.collect(1)
            out = []
            nh = len(_SCALARS) + len(_ARRAYS)
            for r in range(world):
                raw = np.asarray(cells[r])
                head = raw[:8 * nh].view(np.int64)
                one = {k: int(head[j]) for j, k in enumerate(_SCALARS)}
                at = 8 * nh
                for j, (k, dt) in enumerate(_ARRAYS):
                    n = int(head[len(_SCALARS) + j])
                    if n < 0:
                        one[k] = None
                    else:
                        one[k] = raw[at:at + n].copy().view(dt)
                        at += n
                out.append(one)
            ng.release(1)
        dist.barrier()
    finally:
        ng.close()
    return True, out

