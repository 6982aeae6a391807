#!/usr/bin/env python3
"""Turns gpurun_out/<tag>/ (written by tools/profile_bench.sh on the GPU box) into the committed
summaries under profiles/: kernel stats, per-kernel PMC means, and traffic.json (HBM bytes per
scan_perfect_kernel launch, corrected as MI355X_MICROARCH.md's HBM section prescribes).
Usage: python tools/summarize_profile.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_trace.json"), os.path.join(dst, f"{tag}_bench_under_rocprof.json"))

pmc = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        # FETCH_SIZE / WRITE_SIZE are in KB; everything else is a plain count (or a percentage for VALUBusy)
        key = "mean_kb" if c in ("FETCH_SIZE", "WRITE_SIZE") else "mean"
        pmc.setdefault(k, {})[c] = {"launches": len(v), key: sum(v) / len(v)}
with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as fh:
    json.dump(pmc, fh, indent=1, sort_keys=True)

# calibration: calib_stream_read_kernel reads a known byte count (one coalesced dword per lane)
bench = json.load(open(os.path.join(src, "bench_fetch.json")))
calib = pmc["rb::calib_stream_read_kernel"]["FETCH_SIZE"]["mean_kb"] * 1024
events_cap = max(1 << 20, bench["config"]["bases_per_gpu"] // 4)
events_cap = (events_cap + 63) // 64 * 64
known = min(256 << 20, events_cap * 8)
factor = known / calib
scan = pmc["rb::scan_perfect_kernel"]
fetch = scan["FETCH_SIZE"]["mean_kb"] * 1024 * factor
write = scan["WRITE_SIZE"]["mean_kb"] * 1024
out = {
    "tag": tag,
    "fetch_size_correction": factor,
    "calibration": {"kernel": "calib_stream_read_kernel", "known_bytes": known, "FETCH_SIZE_bytes": calib},
    "scan_perfect_kernel_hbm_read_bytes_per_launch": fetch,
    "scan_perfect_kernel_hbm_write_bytes_per_launch": write,
    "scan_perfect_kernel_hbm_bytes_per_launch": fetch + write,
    "algorithmic_read_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    "events_per_launch": bench["device_events_per_step"],
}
for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "VALUBusy", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
    if c in scan:
        out["scan_perfect_kernel_" + c] = scan[c]["mean"]
# the window-stage scan kernels (launched once per pass by --stage-kernels, on the same 100-Mbp record): per base, so
# that bench.py can scale them to the chromosome-sized record of its configs[2] leg
bases = bench["config"]["bases_per_gpu"]
# (the anchored stage runs as two kernels since round 3: the planes kernel scan_anchored_kernel and the window scan of the
# planes scan_xa_window_kernel; "scan_anchored_kernel" below is whichever instantiation ran)
for key, needle in (("scan_window_kernel", "scan_window_kernel<1>"), ("scan_anchored_kernel", "scan_anchored_kernel"), ("scan_xa_window_kernel", "scan_xa_window_kernel")):
    k = [name for name in pmc if needle in name]
    if not k or "FETCH_SIZE" not in pmc[k[0]] or "WRITE_SIZE" not in pmc[k[0]]:
        continue
    m = pmc[k[0]]
    rd, wr = m["FETCH_SIZE"]["mean_kb"] * 1024 * factor, m["WRITE_SIZE"]["mean_kb"] * 1024
    out[key + "_hbm_read_bytes_per_base"] = rd / bases
    out[key + "_hbm_write_bytes_per_base"] = wr / bases
    out[key + "_hbm_bytes_per_base"] = (rd + wr) / bases
    out[key + "_algorithmic_bytes_per_base"] = 0.375
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "VALUBusy", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
        if c in m:
            out[key + "_" + c + ("" if c == "VALUBusy" else "_per_base")] = m[c]["mean"] / (1 if c == "VALUBusy" else bases)
# the anchored stage's scan as a whole = its two kernels
if "scan_xa_window_kernel_hbm_bytes_per_base" in out and "scan_anchored_kernel_hbm_bytes_per_base" in out:
    for q in ("hbm_read_bytes_per_base", "hbm_write_bytes_per_base", "hbm_bytes_per_base", "SQ_INSTS_VALU_per_base", "SQ_INSTS_SALU_per_base"):
        if ("scan_anchored_kernel_" + q) in out and ("scan_xa_window_kernel_" + q) in out:
            out["anchored_stage_scan_" + q] = out["scan_anchored_kernel_" + q] + out["scan_xa_window_kernel_" + q]
    out["anchored_stage_scan_kernels"] = ["scan_anchored_kernel (planes)", "scan_xa_window_kernel"]
with open(os.path.join(dst, "traffic.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out, indent=1))
