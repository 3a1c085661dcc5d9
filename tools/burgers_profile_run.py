#!/usr/bin/env python3
"""The bench's Burgers workload (8192 x 512, 50 sub-steps per launch) and nothing else: the command tools/prof_sq_burgers.sh
profiles.  usage: tools/burgers_profile_run.py [launches]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-pde-control_amd"))
import torch  # noqa: E402
from pdegym.burgers import make_vec  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
E, N = 8192, 512
dev = torch.device("cuda", 0)
env = make_vec(E, config=dict(N=N), device=0)
env.reset(seed=0)
acts = torch.from_numpy(np.random.RandomState(5).uniform(-1, 1, (n, E, 4)).astype(np.float32)).to(dev)
for i in range(n):
    env.step_torch(acts[i])
torch.cuda.synchronize(dev)
assert int(env._status.sum()) == 0
print("ok", E, N, env.cfg_steps, n)
