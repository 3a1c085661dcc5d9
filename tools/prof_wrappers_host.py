import sys, time, cProfile, pstats
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, 'model-based-pde-control_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from pdegym._gym import gym
from pdegym.common import transforms as T
from pdegym.common import vec_wrappers as W

E, N = 1024, 64
class Fake(gym.vector.VectorEnv):
    def __init__(self):
        super().__init__(E, gym.spaces.Box(-np.inf, np.inf, shape=(1, N), dtype=np.float32), gym.spaces.Box(-1.0, 1.0, shape=(1, 4), dtype=np.float32))
        self.o = np.random.RandomState(0).randn(E, 1, N).astype(np.float32); self.t = np.zeros(E, dtype=np.int64)
    def reset(self, **kw): return self.o.copy()
    def step_async(self, a): self.a = a
    def step_wait(self, **kw):
        self.t += 1
        return self.o, np.zeros(E), np.zeros(E, bool), np.zeros(E, bool), {"step": self.t.copy()}
vec = Fake()
ostore = W.StoreNObsVecWrapper(vec, num_steps=1)
e = W.TransformObsWrapper(ostore, T.ScaleTransform(batched=True, aggregate=True, frozen=False), frozen=False)
e = W.TransformObsWrapper(e, T.BatchTransform(T.SensorTransform(stride=1)))
e = W.TransformObsWrapper(e, T.BatchTransform(T.SensorTransform(stride=1)))
astore = W.StoreNActionsVecWrapper(e, num_steps=1)
low, high = vec.single_action_space.low[np.newaxis], vec.single_action_space.high[np.newaxis]
top = W.TransformActionWrapper(astore, T.ScaleTransform(bounds=(low, high), aggregate=True, frozen=True, batched=True).Inverse, frozen=True)
top.reset()
acts = np.random.RandomState(0).uniform(-1, 1, (E, 1, 4)).astype(np.float32)
for _ in range(20): top.step(acts)
t0 = time.perf_counter()
for _ in range(200): top.step(acts)
print("ms/step", (time.perf_counter() - t0) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): top.step(acts)
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(22)
