#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock cycles of cell_fwd_kernel, workgroup 0 (libsurrogate_hip_stamp.so)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-pde-control_amd"))
import torch  # noqa: E402
from pdecontrol.surrogates import hipops  # noqa: E402

hipops.LIB_PATH = os.path.join(ROOT, "model-based-pde-control_amd", "lib", "libsurrogate_hip_stamp.so")
from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch  # noqa: E402

N = 256 if "n256" in sys.argv else 64
dev = torch.device("cuda", 0)
lib = hipops.load()
lib.sur_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
m = build_module(dev, N=N)
batch = synthetic_batch(B=64, N=N, device=dev)
for rep in range(3):
    lib.sur_debug_stamps(None, 1)
    with hipops.inner_forks(False):
        out = m.training_step(batch, 0)
        out["loss"].backward()
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 128)()
    lib.sur_debug_stamps(buf, 0)
print(f"N = {N}: cell_fwd_kernel workgroup 0, both chunks (20 steps: 6 teacher forced), cycles summed")
for i, name in ((109, "prologue (weights, h0/c0)"), (110, "step head, teacher forced (x, h <- lstates (global), c)"),
                (111, "step head, free running (x, h, c from LDS)"), (112, "cell_forward (gates GEMM + activations)"),
                (113, "stores h_all / c_all / saved + barrier"),
                (120, "cell_bwd A: partial sums + gate derivative + dg_all store"), (121, "cell_bwd: issue DMA + prefetch of step k-1"),
                (122, "cell_bwd B: dh GEMM (4 gates on 4 waves) + barrier")):
    print(f"   {name:58s} {buf[i]:8d}")
