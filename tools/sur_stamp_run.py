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
def fwd_bwd():
    for p in m.surrogate.parameters():
        if p.grad is not None:
            p.grad.zero_()
    out = m.training_step(batch, 0)
    out["loss"].backward()
    torch.cuda.synchronize()


fwd_bwd()
lib.sur_debug_stamps(None, 1)
fwd_bwd()
buf = (ctypes.c_longlong * 32)()
lib.sur_debug_stamps(buf, 0)
names = {1: "fwd gates gemm", 2: "fwd gate activations", 3: "fwd deconv0", 4: "fwd LN0+silu", 5: "fwd deconv1",
         6: "fwd LN1+silu", 7: "fwd conv7", 8: "fwd LN2+silu", 9: "fwd conv5", 11: "fwd step output store",
         21: "bwd: recompute forward (incl. the fwd phases above)", 22: "bwd: dd assembly", 23: "bwd: decoder backward",
         24: "bwd: cell elementwise", 25: "bwd: cell GEMMs (dx, dh, gWx, gWh)"}
vals = list(buf)
print("one training step (forward + backward), workgroup 0, shader-clock cycles per rollout step:")
print("(phases 1-9 accumulate over BOTH chunk_fwd and the recompute inside chunk_bwd: 40 executions)")
for i, n in names.items():
    per = vals[i] / (40 if i <= 9 else 20)
    print(f"  {n:55s} {per:9.0f}")
