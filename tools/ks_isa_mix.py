#!/usr/bin/env python3
"""Instruction mix of one RK4 sub-step (the innermost loop body) of the fused KS stepper layouts, from the gfx950
assembly (hipcc -S).  Every wave64 VALU instruction -- fp64 FMA, 32-bit DPP move or select alike -- holds its SIMD's
issue port for 4 cycles, so the VALU count x 4 cycles x sub-steps is the issue floor of a launch with one wave per SIMD.
usage: tools/ks_isa_mix.py > profiles/rNN_ks_isa_mix.txt"""
import collections
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "model-based-pde-control_amd", "csrc", "ks_kernels.hip")
KERNELS = [("C2 default: 1 point/lane, DPP wave chain", "_ZN2ks12ks_rk4_fusedILi1ELi64ELi2ELb0EEEvNS_8StepArgsE", 1),
           ("C2 hybrid: +-1,+-2 DPP, +-3,+-4 ds_bpermute", "_ZN2ks12ks_rk4_fusedILi1ELi64ELi3ELb0EEEvNS_8StepArgsE", 1),
           ("C2 hybrid1: +-1..+-3 DPP, +-4 ds_bpermute", "_ZN2ks12ks_rk4_fusedILi1ELi64ELi4ELb0EEEvNS_8StepArgsE", 1),
           ("C3 default: 16 points/lane, DPP row rotations", "_ZN2ks12ks_rk4_fusedILi16ELi16ELi1ELb0EEEvNS_8StepArgsE", 16),
           ("C3 EXACT mode (reference operation order; reset burn-in)", "_ZN2ks12ks_rk4_fusedILi16ELi16ELi1ELb1EEEvNS_8StepArgsE", 16),
           ("C2 EXACT mode", "_ZN2ks12ks_rk4_fusedILi1ELi64ELi2ELb1EEEvNS_8StepArgsE", 1)]

with tempfile.TemporaryDirectory() as tmp:
    asm = os.path.join(tmp, "ks.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-S",
                    "--cuda-device-only", "-o", asm, SRC], check=True, capture_output=True)
    lines = open(asm).read().splitlines()

for title, sym, ppl in KERNELS:
    start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    # the sub-step loop is the LAST inner loop of the kernel (the first one evaluates phi = actions @ F)
    heads = [i for i, l in enumerate(body) if "Loop Header" in l]
    lo = heads[-1]
    hi = next(i for i in range(lo, len(body)) if "s_cbranch" in body[i])
    ops = collections.Counter()
    for l in body[lo + 1:hi + 1]:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        ops[t.split()[0]] += 1
    valu = sum(n for k, n in ops.items() if k.startswith("v_"))
    lds = sum(n for k, n in ops.items() if k.startswith("ds_"))
    fp64 = sum(n for k, n in ops.items() if k.startswith("v_") and "f64" in k)
    print(f"== {title}")
    print(f"   VALU {valu} per lane and sub-step ({valu / ppl:.1f} per grid point), of which fp64 arithmetic {fp64}, "
          f"DPP moves {ops.get('v_mov_b32_dpp', 0)}, selects {ops.get('v_cndmask_b32_e32', 0)}; LDS-pipe (ds_bpermute) {lds}; "
          f"s_nop {ops.get('s_nop', 0)}, s_waitcnt {ops.get('s_waitcnt', 0)}")
    print(f"   issue floor at one wave per SIMD: {valu} x 4 cycles = {valu * 4} cycles per sub-step per wave")
    for k, n in ops.most_common():
        print(f"      {n:5d}  {k}")
