#!/usr/bin/env python3
"""Register / LDS / scratch footprint of every kernel this repo compiles for gfx950, from the assembler metadata
(hipcc -S --cuda-device-only), and the wave occupancy it allows: gfx950 has 512 unified VGPRs per SIMD lane, allocated
in granules of 8, at most 8 waves per SIMD.  Dynamic LDS is chosen at launch and is not in this table (DESIGN.md lists
the per-kernel sizes).  usage: tools/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "model-based-pde-control_amd", "csrc")
SOURCES = [("ks_kernels.hip", ["-ffp-contract=off"]), ("sur_kernels.hip", []), ("burgers.hip", ["-ffp-contract=off"]),
           ("spectral.hip", [])]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout
        return out.strip().splitlines()
    except Exception:
        return names


print("%-92s %5s %5s %5s %7s %8s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "lds(st)", "waves"))
for src, extra in SOURCES:
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", asm,
                        os.path.join(CSRC, src)] + extra, check=True, capture_output=True)
        text = open(asm).read()
    rows = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", text, re.S):
        body = m.group(2)
        get = lambda key, d=0: int((re.search(r"\.amdhsa_%s (\d+)" % key, body) or [None, d])[1])
        total = get("next_free_vgpr")
        accum = get("accum_offset", total)
        rows.append((m.group(1), min(accum, total), max(total - accum, 0), get("next_free_sgpr"), get("private_segment_fixed_size"),
                     get("group_segment_fixed_size"), total))
    names = demangle([r[0] for r in rows])
    print(f"# {src}")
    for (raw, vgpr, agpr, sgpr, scratch, lds, total), name in zip(rows, names):
        alloc = (max(total, 1) + 7) // 8 * 8
        waves = min(8, 512 // alloc)
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*$", "", name)              # drop the argument list
        if src == "ks_kernels.hip" and "ks_rk4_fused<" in name and not re.search(
                r"<(1, 64, [234]|4, 16, 1|16, 16, 1), (true|false)>", name):
            continue                                    # the layouts the chooser picks for the BASELINE configs + the hybrids
        print("%-92s %5d %5d %5d %7d %8d %6d" % (name[:92], vgpr, agpr, sgpr, scratch, lds, waves))
