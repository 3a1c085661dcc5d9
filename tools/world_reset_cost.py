#!/usr/bin/env python3
"""f2: where a device-resident imagined-rollout step goes -- steps without resets, the reset alone, its parts."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "model-based-pde-control_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pdegym  # noqa: E402,F401
import _world_scenario  # noqa: E402,F401
from test_world_env import namespace  # noqa: E402

dev = torch.device("cuda", 0)
M = namespace()
env = M.Env()
tstep, tau = env.cfg_steps * env.dt, 5
rpw = M.Replay()
r = np.random.RandomState(5)
for ep in range(6):
    for t in range(40):
        rpw.add([M.Sample(r.randn(1, 64).astype(np.float32), r.uniform(-1, 1, (1, 4)).astype(np.float32),
                          r.randn(1, 64).astype(np.float32), np.float32(0), False, t == 39, np.int32(t + 1))])
forcing = M.T.BatchTransform(env.forcing)
stransf = M.T.SampleTransform(None, [forcing])
mods = []
for seed in range(3):
    torch.manual_seed(seed)
    f = M.factory_cls()
    mods.append(M.TrainingModule(surrogate=f.surrogate(delta=tstep, dscaling=None, tau=tau, **f.model()),
                                 loss=torch.nn.MSELoss(reduction="none"), tstep=tstep, delta=tstep, tau=tau, tbtt=10).to(dev))
world = M.WorldVecEnv(surrogate=M.Ensemble(mods), observation_space=env.observation_space, action_space=env.action_space,
                      max_episode_steps=100000, stransf=stransf.Inverse, reward_func=env.reward_func, num_envs=100, horizon=10 ** 9,
                      tstep=tstep, batched_reward_func=env.batched_reward_func, device_resident=True)
world.setup(M.ds.StartingStateDataset(data=rpw.data, length=tau, stride=1, bootstrapping=False, stransf=stransf))
world.reset()
acts = r.uniform(-1, 1, (100, 1, 64)).astype(np.float32)
for _ in range(5):
    world.step(acts)
t0 = time.perf_counter()
for _ in range(100):
    world.step(acts)
torch.cuda.synchronize()
print(f"step without reset: {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms")
for _ in range(3):
    world.reset()
t0 = time.perf_counter()
for _ in range(30):
    world.reset()
torch.cuda.synchronize()
print(f"reset: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms")
if world._dev_starting is not None:
    t0 = time.perf_counter()
    for _ in range(30):
        world._dev_starting.next_batch()
    torch.cuda.synchronize()
    print(f"  of which next_batch (index rule + gathers + transforms): {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms")
if "cprofile" in sys.argv:
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        world.step(acts)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
