#!/usr/bin/env python3
"""A/B of the captured TBPTT step: ``python tools/tbptt_ab.py [reps]`` times GraphedTBPTTStep with and without the
chunk pipeline (hipops.fused_tbptt_train) at N = 64 and N = 256, B = 64, alternating the arms (same process, same box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-pde-control_amd"))
import torch  # noqa: E402
from pdecontrol.surrogates import hipops  # noqa: E402
if os.environ.get("SUR_LIB"):       # A/B of kernel builds: SUR_LIB=<path to a libsurrogate_hip variant>
    hipops.LIB_PATH = os.path.abspath(os.environ["SUR_LIB"])
from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch  # noqa: E402
from pdecontrol.surrogates.graph_step import GraphedTBPTTStep  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)


def timed(g, steps=300):
    for _ in range(30):
        g.step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        g.step()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps * 1e3


for N in (64, 256):
    batch = synthetic_batch(B=64, N=N, device=dev)
    arms = {}
    for name, flag in (("pipelined", True), ("combined", False)):
        g = GraphedTBPTTStep(build_module(dev, N=N), tuple(batch[0].shape), pipelined=flag)
        g.step(*batch)
        arms[name] = g
    for r in range(reps):
        row = {name: timed(g) for name, g in arms.items()}
        print(f"N={N} rep {r}: " + "  ".join(f"{k} {v:.4f} ms" for k, v in row.items()) +
              "  loss " + " / ".join(f"{float(g.result['loss']):.6f}" for g in arms.values()), flush=True)
