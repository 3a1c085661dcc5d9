#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock shares of chunk_fwd (libsurrogate_hip_stamp.so, -DSUR_STAMP build)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-pde-control_amd"))
import torch  # noqa: E402
from pdecontrol.surrogates import hipops, ops  # noqa: E402

hipops.LIB_PATH = os.path.join(ROOT, "model-based-pde-control_amd", "lib", "libsurrogate_hip_stamp.so")
from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch  # noqa: E402

dev = torch.device("cuda", 0)
ops.enable_fused(True)
lib = hipops.load()
lib.sur_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
m = build_module(dev)
batch = synthetic_batch(B=64, device=dev)
with torch.no_grad():
    m.training_step(batch, 0)
    torch.cuda.synchronize()
    lib.sur_debug_stamps(None, 1)
    m.training_step(batch, 0)
    torch.cuda.synchronize()
buf = (ctypes.c_longlong * 32)()
lib.sur_debug_stamps(buf, 0)
names = ["(gap)", "gates gemm x8", "gate activations", "deconv0", "LN0+silu", "deconv1", "LN1+silu", "conv7", "LN2+silu",
         "conv5", "step input load", "step output store"]
vals = list(buf)[:12]
tot = sum(vals[1:])
print("20 steps, workgroup 0, cycles (shader clock):")
for n, v in zip(names, vals):
    print(f"  {n:22s} {v:10d}  {100.0 * v / max(tot, 1):5.1f}%   per step {v / 20:9.0f}")
print("total per step", tot / 20)
