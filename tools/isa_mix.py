#!/usr/bin/env python3
"""Instruction mix of scan_perfect_kernel: which share of its executed VALU instructions is v_alignbit_b32 (the funnel
shift that issues at ~0.6x the rate of the other integer VALU ops, profiles/r01b_valu_peak_probe.txt).

No counter sees opcodes, so the share is put together from two things that ARE measured or exact:
  * dynamic weights: SQ_INSTS_VALU per launch of the product build and of two builds with parts compiled out
    (tools/build_variant.sh + tools/variant_pmc.sh on the GPU box): base = Z + prefilter + loop, chain = doubling chain +
    candidate queue, staging = event staging;
  * static content of branch-free blocks, from the ISA of the built kernel (this script compiles kernels.hip to
    assembly): the Z block runs once per (wavefront, motif) pair and holds exactly 20 v_alignbit; a run of the chain holds
    61 of its ~181 VALU; the staging code holds none.
The chain's weight also contains the queue's bookkeeping VALU (no alignbit): the estimate prices the whole weight at the
chain's own alignbit density, the bracket runs from "no alignbit in the chain part" to "all of it as dense as the densest block".
Usage: python tools/isa_mix.py [gpurun_out/variant_pmc]  ->  profiles/isa_mix.json"""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "ribbit_amd", "csrc")


def kernel_blocks():
    """basic blocks of scan_perfect_kernel: [(label, n_valu, n_alignbit, has_branch_inside)]"""
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "kernels.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{SRC}", "-S",
                        "--cuda-device-only", "-o", asm, os.path.join(SRC, "kernels.hip")], check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(text) if re.match(r"^_ZN2rb19scan_perfect_kernel.*:", l))
    end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
    blocks, cur = [], ["entry", 0, 0]
    for l in text[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(tuple(cur))
            cur = [m.group(1), 0, 0]
            continue
        op = l.strip().split()[0] if l.strip() and not l.strip().startswith((";", ".")) else ""
        if op.startswith("v_"):
            cur[1] += 1
            cur[2] += op.startswith("v_alignbit")
    blocks.append(tuple(cur))
    return blocks


def pmc_valu(directory, variant):
    f = glob.glob(os.path.join(directory, variant, "*", "*_counter_collection.csv"))[0]
    launches = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "scan_perfect_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_INSTS_VALU":
            launches[r["Dispatch_Id"]] = float(r["Counter_Value"])
    return list(launches.values())[0]          # first record of perfect_probe.py = the bench's simulated record


if __name__ == "__main__":
    pmc_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "variant_pmc")
    blocks = kernel_blocks()
    # the Z block: the only block with 20 alignbit among ~40 VALU; chain blocks: those with >= 40 alignbit
    z = [b for b in blocks if b[2] == 20 and b[1] <= 48]
    chains = [b for b in blocks if b[2] >= 40]
    assert len(z) == 1, z
    bases, tile = 100_000_000, 64 * 8 * 32
    pairs = -(-bases // tile) * 99
    total, nostage, nochain = (pmc_valu(pmc_dir, v) for v in ("product", "nostage", "nochain"))
    # a run of the chain executes its blocks whole (in place: two blocks; from the queue: one): alignbit / VALU over all of them.
    # The loosest bound takes the densest single block instead.
    chain_all = [b for b in blocks if b[2] >= 10 and b is not z[0]]
    chain_share = sum(b[2] for b in chain_all) / sum(b[1] for b in chain_all)
    chain_share_max = max(b[2] / b[1] for b in chains)
    align_base = z[0][2] * pairs
    align_chain = chain_share * (nostage - nochain)
    align_chain_max = chain_share_max * (nostage - nochain)
    out = {
        "kernel": "scan_perfect_kernel", "record": "bench.py's synthetic FASTA, 100 Mbp, m = 2..100",
        "method": "dynamic weights from SQ_INSTS_VALU of the product and two ablated builds (tools/variant_pmc.sh); v_alignbit content "
                  "of branch-free ISA blocks (tools/isa_mix.py); the chain part is priced at the chain's own alignbit density (its weight "
                  "also holds the candidate queue's bookkeeping), bracket = [none, densest block]",
        "valu_per_launch": {"product": total, "base_Z_prefilter_loop": nochain, "chain_and_queue": nostage - nochain, "event_staging": total - nostage},
        "wavefront_motif_pairs": pairs, "valu_per_pair_base": nochain / pairs,
        "isa": {"z_block": {"label": z[0][0], "valu": z[0][1], "alignbit": z[0][2]},
                "chain_blocks": [{"label": b[0], "valu": b[1], "alignbit": b[2]} for b in chain_all],
                "chain_alignbit_per_valu": chain_share, "densest_chain_block": chain_share_max},
        "alignbit_per_launch": {"base_exact": align_base, "chain_estimate": align_chain, "chain_loosest_bound": align_chain_max, "staging": 0},
        "alignbit_share": (align_base + align_chain) / total,
        "alignbit_share_bracket": [align_base / total, (align_base + align_chain_max) / total],
    }
    path = os.path.join(ROOT, "profiles", "isa_mix.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("valu_per_launch", "valu_per_pair_base", "alignbit_per_launch", "alignbit_share", "alignbit_share_bracket")}, indent=1))
