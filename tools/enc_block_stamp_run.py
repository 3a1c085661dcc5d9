#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock cycles of enc_block_bwd_multi_kernel, workgroup 0 (first action-encoder sample), one
residual block per launch (libsurrogate_hip_stamp.so, `make -C model-based-pde-control_amd/csrc stamp`).  ``n256`` for N = 256."""
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
names = ["entry", "stage_weights+setup_grads", "loads (in, a1pre a1 a2pre, s, dout)", "LN3 bwd", "skip: wgrad<1> || dgrad<1>", "LN2+SiLU bwd",
         "conv2: wgrad<3> || dgrad<3>", "LN1+SiLU bwd", "conv1: wgrad<3> || dgrad<3>", "store din / loop end", "add_to_row"]
for rep in range(3):
    lib.sur_debug_stamps(None, 1)
    out = m.training_step(batch, 0)
    out["loss"].backward()
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 128)()
    lib.sur_debug_stamps(buf, 0)
print(f"N = {N}: workgroup 0 of enc_block_bwd_multi (action-encoder job), cycles per phase (stamps add a barrier each)")
for blk in (2, 1, 0):
    base = 64 + 16 * blk
    tot = sum(buf[base + i] for i in range(1, 11))
    print(f" block {blk}: {tot} cycles after entry")
    for i in range(1, 11):
        print(f"   {names[i]:40s} {buf[base + i]:8d}")
