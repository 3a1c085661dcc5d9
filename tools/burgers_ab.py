#!/usr/bin/env python3
"""Timing of the Burgers stepper (BASELINE configs[4] shape and a few others); ``BG_LIB=<path>`` selects a library build."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-pde-control_amd"))
import torch  # noqa: E402
from pdegym.burgers import _hip  # noqa: E402
if os.environ.get("BG_LIB"):
    _hip.LIB_PATH = os.path.abspath(os.environ["BG_LIB"])
from pdegym.burgers import make_vec  # noqa: E402

dev = torch.device("cuda", 0)
for E, N in ((8192, 512), (8192, 128), (32768, 512), (1024, 64)):
    env = make_vec(E, config=dict(N=N), device=0)
    env.reset(seed=0)
    acts = torch.from_numpy(np.random.RandomState(5).uniform(-1, 1, (25, E, 4)).astype(np.float32)).to(dev)
    for i in range(5):
        env.step_torch(acts[i])
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for i in range(20):
        ev[i][0].record()
        env.step_torch(acts[5 + i])
        ev[i][1].record()
    torch.cuda.synchronize(dev)
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    print(f"E={E} N={N}: {ms:.4f} ms per launch, {E * env.cfg_steps / (ms * 1e-3):.3e} sub-steps/s, checksum {float(env.u.double().sum()):.9e}",
          flush=True)
