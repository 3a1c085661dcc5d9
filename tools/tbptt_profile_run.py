#!/usr/bin/env python3
"""Runs N fused (or plain) HIP-graph TBPTT steps; meant to be wrapped by rocprofv3 --kernel-trace --stats."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-pde-control_amd"))
import torch  # noqa: E402
from pdecontrol.surrogates import ops  # noqa: E402
from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch  # noqa: E402
from pdecontrol.surrogates.graph_step import GraphedTBPTTStep  # noqa: E402

fused = "plain" not in sys.argv[1:]
steps = 20
dev = torch.device("cuda", 0)
ops.enable_fused(fused)
batch = synthetic_batch(B=64, device=dev)
g = GraphedTBPTTStep(build_module(dev), tuple(batch[0].shape))
g.step(*batch)
for _ in range(steps):
    g.step()
torch.cuda.synchronize()
print("fused" if fused else "plain", "loss", float(g.result["loss"].detach()))
