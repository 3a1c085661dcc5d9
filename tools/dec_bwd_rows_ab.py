#!/usr/bin/env python3
"""Isolated (no side streams) kernel durations of the eager fused N = 256 backward against the number of partial rows
(= workgroups of the pair-parallel kernels): run under rocprofv3 --kernel-trace --stats.  usage: dec_bwd_rows_ab.py <rows>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-pde-control_amd"))
os.environ["PDECONTROL_SPLIT_GRAPHS"] = "0"
import torch  # noqa: E402
from pdecontrol.surrogates import hipops, ops  # noqa: E402
from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch  # noqa: E402

hipops.CHUNK_ROWS = int(sys.argv[1])
dev = torch.device("cuda", 0)
ops.enable_fused(True)
m = build_module(dev, N=256)
batch = synthetic_batch(B=64, N=256, device=dev)
with hipops.inner_forks(False):
    for _ in range(6):
        for p in m.surrogate.parameters():
            p.grad = None
        out = m.training_step(batch, 0)
        out["loss"].backward()
    torch.cuda.synchronize()
print("rows", hipops.CHUNK_ROWS, "loss", float(out["loss"]))
