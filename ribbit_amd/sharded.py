"""Chunk-sharded scan of ONE long record over several GPUs (BASELINE.json north_star; SURVEY.md 8e
option 2).  Every rank loads its chunk plus halos as a record of its own, runs the three scan kernels,
keeps the events it owns, and the ranks send events, packed planes and composed planes to the ONE rank that
runs the host merge (gather-v over RCCL: torch.distributed "nccl"; "gloo" in the CPU tests) -- nothing is
replicated on the other ranks.  The order-dependent host replay (pairing, window state machines, seed
merges) then runs once on the gathered data, exactly as for a single GPU.  Correctness does not depend on the partition: events are local functions of the
sequence, and the halos cover their reach (include/ribbit_hip.h, "chunk-sharded operation")."""
from __future__ import annotations

import numpy as np

import ribbit_amd


def plan_chunks(length: int, nparts: int, max_motif: int):
    """[(own_lo, own_hi, load_lo, load_hi)] in record coordinates; own ranges are 32-aligned (the composed
    planes are exchanged as whole words) and the last one includes position `length` (end-of-record events)."""
    s = max_motif + 2
    halo_left, halo_right = 2 * s + 64, 4 * s + 64
    cuts = [(length * k // nparts) // 32 * 32 for k in range(nparts)] + [length + 1]
    plans = []
    for k in range(nparts):
        own_lo, own_hi = cuts[k], cuts[k + 1]
        load_lo = max(0, (own_lo - halo_left) // 32 * 32)
        load_hi = min(length, own_hi + halo_right)
        plans.append((own_lo, own_hi, load_lo, load_hi))
    return plans


def scan_part(scanner: "ribbit_amd.Scanner", sequence: bytes, plan, anchored: bool = True) -> dict:
    """What one rank contributes: its own events of the three stages and its own words of all planes."""
    own_lo, own_hi, load_lo, load_hi = plan
    scanner.load_record(sequence[load_lo:load_hi])
    part = {"own_lo": own_lo, "own_hi": own_hi}
    for stage in (0, 1, 2):
        if stage == 2 and not anchored:
            part["ev2"] = part["cnt2"] = None
            continue
        ev, cnt = scanner.stage_events(stage, own_lo - load_lo, own_hi - load_lo, load_lo)
        part[f"ev{stage}"], part[f"cnt{stage}"] = ev, cnt
    w_lo = (own_lo - load_lo) // 32
    w_hi = min((own_hi - load_lo + 31) // 32, (load_hi - load_lo) // 32 + 1)
    for which, key in ((0, "hi"), (1, "lo"), (2, "brk")):
        part[key] = scanner.packed_plane(which)[w_lo:w_hi].copy()
    part["xa"] = scanner.xa_words(w_lo, w_hi) if anchored else None
    return part


def merge_parts(min_motif: int, max_motif: int, length: int, parts: list) -> dict:
    """Host replay on the union of the parts (rank 0 after the exchange)."""
    nwords = length // 32 + 1 + (max_motif + 2) // 32 + 4
    stride = (length // 32 + 1 + 7) // 8 * 8 + 16
    hi = np.zeros(nwords, "<u4"); lo = np.zeros(nwords, "<u4"); brk = np.full(nwords, 0xFFFFFFFF, "<u4")
    nm = max_motif - min_motif + 1
    anchored = all(p.get("xa") is not None for p in parts)
    xa = np.zeros((nm, stride), "<u4") if anchored else None
    for p in parts:
        w0 = p["own_lo"] // 32
        n = len(p["hi"])
        hi[w0:w0 + n], lo[w0:w0 + n], brk[w0:w0 + n] = p["hi"], p["lo"], p["brk"]
        if anchored:
            xa[:, w0:w0 + n] = p["xa"]
    return ribbit_amd.host_scan_from_events(min_motif, max_motif, length, hi, lo, brk,
                                            np.ascontiguousarray(xa) if anchored else None, stride, parts)


def gather_parts(part: dict, device=None, dst: int = 0):
    """gather-v of every array of `part` to rank `dst` (torch.distributed must be initialised): the list of all ranks'
    parts on dst, None elsewhere.  The composed planes are 12.4 B/base at 99 motif sizes: only dst receives them."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    keys = ["ev0", "cnt0", "ev1", "cnt1", "ev2", "cnt2", "hi", "lo", "brk", "xa"]
    meta = torch.tensor([part["own_lo"], part["own_hi"]] +
                        [(-1 if part.get(k) is None else part[k].size * part[k].itemsize) for k in keys], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    metas = [m.cpu().numpy() for m in metas]
    out = [{"own_lo": int(m[0]), "own_hi": int(m[1])} for m in metas]
    dtypes = {"hi": "<u4", "lo": "<u4", "brk": "<u4", "xa": "<u4"}
    for j, k in enumerate(keys):
        sizes = [int(m[2 + j]) for m in metas]
        if any(sz < 0 for sz in sizes):
            for o in out:
                o[k] = None
            continue
        cap = max(max(sizes), 1)
        buf = torch.zeros(cap, dtype=torch.uint8, device=dev)
        mine = part[k]
        if mine.size:
            buf[:mine.size * mine.itemsize] = torch.from_numpy(np.ascontiguousarray(mine).view(np.uint8).reshape(-1)).to(dev)
        gathered = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
        dist.gather(buf, gathered, dst=dst)
        if rank != dst:
            continue
        for r in range(world):
            raw = gathered[r][:sizes[r]].cpu().numpy()
            out[r][k] = raw.view(dtypes.get(k, "<u8")).copy()
    if rank != dst:
        return None
    nm = len(out[0]["cnt0"])
    for o in out:
        if o["xa"] is not None:
            o["xa"] = o["xa"].reshape(nm, -1)
    return out
