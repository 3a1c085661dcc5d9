#!/usr/bin/env python3
"""Runs TBPTT steps for a profiler: ``[plain] [eager] [n256]`` -- default: fused kernels, captured graph, N = 64.
Wrapped by rocprofv3 --kernel-trace --stats (tools/prof_tbptt.sh); ``eager cprofile`` prints the host-side profile of
the eager fused step instead (where the Python time of the path pl.Trainer.fit drives goes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-pde-control_amd"))
import torch  # noqa: E402
from pdecontrol.surrogates import ops  # noqa: E402
from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch, time_eager  # noqa: E402
from pdecontrol.surrogates.graph_step import GraphedTBPTTStep  # noqa: E402

args = sys.argv[1:]
fused = "plain" not in args
N = 256 if "n256" in args else 64
steps = 20
dev = torch.device("cuda", 0)
ops.enable_fused(fused)
batch = synthetic_batch(B=64, N=N, device=dev)
if "eager" in args:
    module = build_module(dev, N=N)
    time_eager(module, batch, steps=5, warmup=3)
    if "cprofile" in args:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        dt, loss = time_eager(module, batch, steps=100, warmup=0)
        pr.disable()
        print(f"eager fused N={N}: {dt * 1e3:.3f} ms/step under cProfile")
        pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
    else:
        dt, loss = time_eager(module, batch, steps=steps, warmup=0)
    print("eager", "fused" if fused else "plain", f"N={N}", "ms/step", dt * 1e3, "loss", loss)
else:
    g = GraphedTBPTTStep(build_module(dev, N=N), tuple(batch[0].shape))
    g.step(*batch)
    for _ in range(steps):
        g.step()
    torch.cuda.synchronize()
    print("fused" if fused else "plain", f"N={N}", "loss", float(g.result["loss"]))
