#!/usr/bin/env python3
"""Phase times of fno_forward_kernel from shader-clock stamps (diagnostic build: make -C csrc fno-stamp).  GPU box only."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-pde-control_amd"))
import torch  # noqa: E402
from pdecontrol.surrogates import spectral  # noqa: E402
spectral.LIB_PATH = os.path.join(ROOT, "model-based-pde-control_amd", "lib", "libspectral_hip_stamp.so")
from pdecontrol.architectures import BurgersFNO  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
f = BurgersFNO()
sur = f.surrogate(delta=0.05, dscaling=None, tau=5, **f.model()).to(dev)
g = torch.Generator().manual_seed(1)
B, N = 64, 512
st, ac = (torch.rand(B, 1, 1, N, generator=g) * 2 - 1).to(dev), (torch.rand(B, 1, 1, N, generator=g) * 2 - 1).to(dev)
for _ in range(3):
    r = sur.rollout(states=st, actions=ac, times=torch.zeros(1), targets=torch.full((1,), 0.05))
torch.cuda.synchronize()
lib = spectral.load()
buf = (ctypes.c_ulonglong * 32)()
assert lib.fno_read_stamps(buf, 32) == 0
t = list(buf)
spans = [("cos + lift", 0, 1), ("table", 1, 2)]
for l in range(4):
    spans += [(f"L{l} A dft", 3 + 4 * l, 4 + 4 * l), (f"L{l} B mix", 4 + 4 * l, 5 + 4 * l), (f"L{l} C idft+pw", 5 + 4 * l, 6 + 4 * l)]
spans.append(("project", 19, 20))
tot = t[20] - t[0]
print(f"forward kernel, workgroup 0: {tot} shader clocks")
for name, a, b in spans:
    print(f"  {name:16s} {t[b] - t[a]:8d}  {100.0 * (t[b] - t[a]) / tot:5.1f} %")
