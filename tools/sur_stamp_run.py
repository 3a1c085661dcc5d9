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
N = 256 if "n256" in sys.argv[1:] else 64
m = build_module(dev, N=N)
batch = synthetic_batch(B=64, N=N, device=dev)
def fwd_bwd():
    with hipops.inner_forks(False):   # one kernel at a time: the stamp buffer is shared by every launch
        _fwd_bwd()


def _fwd_bwd():
    for p in m.surrogate.parameters():
        if p.grad is not None:
            p.grad.zero_()
    out = m.training_step(batch, 0)
    out["loss"].backward()
    torch.cuda.synchronize()


fwd_bwd()
lib.sur_debug_stamps(None, 1)
fwd_bwd()
buf = (ctypes.c_longlong * 128)()      # the library copies all 128 slots
lib.sur_debug_stamps(buf, 0)
names = {1: "cell fwd: gates GEMM", 2: "cell fwd: gate activations", 3: "dec fwd: deconv0", 4: "dec fwd: LN0+silu",
         5: "dec fwd: deconv1", 6: "dec fwd: LN1+silu", 7: "dec fwd: conv7", 8: "dec fwd: LN2+silu", 9: "dec fwd: conv5",
         12: "dec bwd: conv5 weight grad", 13: "dec bwd: conv5 data grad", 14: "dec bwd: LN2 bwd",
         15: "dec bwd: conv7 weight grad", 16: "dec bwd: conv7 data grad", 17: "dec bwd: LN1 bwd",
         18: "dec bwd: deconv1 weight grad", 19: "dec bwd: deconv1 data grad", 26: "dec bwd: LN0 bwd",
         27: "dec bwd: deconv0 weight grad"}
vals = list(buf)
print(f"N = {N}: one training step (forward + backward), workgroup 0 of every launch, shader-clock cycles summed over the step:")
print("(cell phases: 20 executions (2 chunks x 10 steps); decoder phases: the (step, sample) pairs workgroup 0 handles;")
print(" the first phase of each kernel also contains the time since the previous stamped kernel ended)")
for i, n in sorted(names.items()):
    print(f"  {n:55s} {vals[i]:9d}")
print("encoder block backward (enc_block_bwd_multi_kernel, workgroup 0 = first sample of the first wide job = the state encoder;")
print(" summed over the launches of the step -- one per chunk in the un-forked schedule):")
phases = ["entry (since the previous stamp: meaningless)", "stage weights, zero the accumulators", "load input / records / dout",
          "LayerNorm 3 backward", "skip: weight | data gradient", "LayerNorm 2 backward", "conv 2: weight | data gradient",
          "LayerNorm 1 backward", "conv 1: weight | data gradient", "store din, loop end", "add the accumulators to the partial row"]
for blk in (2, 1, 0):
    for k, n in enumerate(phases):
        if k:
            print(f"  block {blk}: {n:45s} {vals[64 + 16 * blk + k]:9d}")
