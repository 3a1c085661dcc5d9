#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock cycles of enc_bwd_kernel, workgroup 0 (libsurrogate_hip_stamp.so, `make stamp`)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-pde-control_amd"))
import torch  # noqa: E402
from pdecontrol.surrogates import hipops, ops  # noqa: E402

hipops.LIB_PATH = os.path.join(ROOT, "model-based-pde-control_amd", "lib", "libsurrogate_hip_stamp.so")
from pdecontrol.surrogates.bench_tbptt import build_module  # noqa: E402

dev = torch.device("cuda", 0)
ops.enable_fused(True)
lib = hipops.load()
lib.sur_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
m = build_module(dev)
owner = hipops.packs_for(m.surrogate, 64, 64)
names = {32: "entry (since previous stamp: meaningless)", 33: "stage_weights", 34: "setup_grads (zero LDS accumulators)",
         35: "load x", 36: "forward recompute (3 blocks)", 37: "load dz", 40: "backward block 2", 39: "backward block 1",
         38: "backward block 0", 41: "store dx / loop end", 42: "epilogue: add LDS accumulators to the partial row"}
for samples in (64, 256, 1280):
    x = torch.rand(samples, 1, 64, device=dev).requires_grad_(False)
    for rep in range(2):
        lib.sur_debug_stamps(None, 1)
        z = hipops.encode(x, owner.action_enc, owner)
        z.sum().backward()
        torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 128)()      # the library copies all 128 slots
    lib.sur_debug_stamps(buf, 0)
    iters = max(1, samples // 256)
    print(f"--- {samples} samples ({iters} per workgroup): cycles (per-sample phases are totals over {iters} iterations)")
    for i in sorted(names):
        print(f"  {names[i]:55s} {buf[i]:9d}")
