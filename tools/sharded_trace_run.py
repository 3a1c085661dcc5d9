#!/usr/bin/env python3
"""KSShardedVecEnv steps, host time split into step_async / step_wait (also the command to wrap in rocprofv3 --kernel-trace).
usage: sharded_trace_run.py <E> <N> <L> <handles>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-pde-control_amd"))
import numpy as np
from pdegym.kuramoto import make_vec
E, N, L, H = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
env = make_vec(E, config=dict(L=L, N=N), burn_in=False, devices=[0] * H)
env.reset(seed=0)
acts = np.random.RandomState(0).uniform(-1, 1, (30, E, 1, 4)).astype(np.float32)
for i in range(5):
    env.step(acts[i])
ta = tw = 0.0
t0 = time.perf_counter()
for i in range(5, 30):
    a = time.perf_counter()
    env.step_async(acts[i])
    b = time.perf_counter()
    env.step_wait()
    c = time.perf_counter()
    ta += b - a
    tw += c - b
dt = (time.perf_counter() - t0) / 25
print(f"handles {H}: ms per step {dt * 1e3:.3f}  (step_async {ta / 25 * 1e3:.3f}, step_wait {tw / 25 * 1e3:.3f})", flush=True)
