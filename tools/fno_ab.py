#!/usr/bin/env python3
"""A/B of builds of libspectral_hip.so on the FNO TBPTT step (B = 64, T = 20, N = 512: bench.py's ``fno_tbptt`` leg):
``SPECTRAL_LIB=<path> python tools/fno_ab.py`` times the eager step (one launch per model evaluation and direction)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-pde-control_amd"))
import torch  # noqa: E402
from pdecontrol.surrogates import spectral  # noqa: E402
if os.environ.get("SPECTRAL_LIB"):
    spectral.LIB_PATH = os.path.abspath(os.environ["SPECTRAL_LIB"])
from pdecontrol.architectures import BurgersFNO  # noqa: E402
from pdecontrol.surrogates.training import PDETrainingModule  # noqa: E402

dev = torch.device("cuda", 0)
N = 512
torch.manual_seed(0)
f = BurgersFNO()
sur = f.surrogate(delta=0.05, dscaling=None, tau=5, **f.model())
mod = PDETrainingModule(surrogate=sur, loss=torch.nn.MSELoss(reduction="none"), tstep=0.05, delta=0.05, tau=5, tbtt=10).to(dev)
g = torch.Generator().manual_seed(1)
batch = ((torch.rand(64, 20, 1, N, generator=g) * 2 - 1).to(dev), (torch.rand(64, 20, 1, N, generator=g) * 2 - 1).to(dev))
opt = mod.configure_optimizers()[0][0]


def one():
    o = mod.training_step(batch, 0)
    opt.zero_grad(set_to_none=True)
    o["loss"].backward()
    opt.step()
    return o


for _ in range(5):
    o = one()
torch.cuda.synchronize(dev)
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        o = one()
    torch.cuda.synchronize(dev)
    print(f"{spectral.LIB_PATH.split('/')[-1]} rep {rep}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step  loss {float(o['loss'].detach()):.6f}", flush=True)
