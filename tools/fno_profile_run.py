#!/usr/bin/env python3
"""Eager FNO TBPTT steps for rocprofv3 --kernel-trace --stats (tools/prof_fno.sh) or, with `cprofile`, the host profile."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-pde-control_amd"))
import torch  # noqa: E402
from pdecontrol.architectures import BurgersFNO  # noqa: E402
from pdecontrol.surrogates.training import PDETrainingModule  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
f = BurgersFNO()
sur = f.surrogate(delta=0.05, dscaling=None, tau=5, **f.model())
mod = PDETrainingModule(surrogate=sur, loss=torch.nn.MSELoss(reduction="none"), tstep=0.05, delta=0.05, tau=5, tbtt=10).to(dev)
g = torch.Generator().manual_seed(1)
N = 512
batch = ((torch.rand(64, 20, 1, N, generator=g) * 2 - 1).to(dev), (torch.rand(64, 20, 1, N, generator=g) * 2 - 1).to(dev))
opt = mod.configure_optimizers()[0][0]


def one():
    o = mod.training_step(batch, 0)
    opt.zero_grad(set_to_none=True)
    o["loss"].backward()
    opt.step()


for _ in range(3):
    one()
torch.cuda.synchronize()
if "cprofile" in sys.argv:
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        one()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
else:
    t0 = time.perf_counter()
    for _ in range(5):
        one()
    torch.cuda.synchronize()
    print("eager ms/step", (time.perf_counter() - t0) / 5 * 1e3)
    if "graph" in sys.argv:      # the same step captured as one hipGraph (PDETrainingModule.fused_step)
        for _ in range(3):
            mod.fused_step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            mod.fused_step(batch)
        torch.cuda.synchronize()
        print("graphed ms/step", (time.perf_counter() - t0) / 20 * 1e3)
