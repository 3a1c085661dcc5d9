"""Viscous-Burgers control environments on the MI355X HIP stepper (BASELINE configs[4]).

There is nothing to mirror: the reference imports ``pdegym.burgers`` (pdegym/__init__.py:2) but ships no such module.
What the reference does fix is the discretisation -- ``BurgersPhyPDELoss`` (pdecontrol/surrogates/phyloss/phyloss.py:36-86):
``u_t = nu u_xx - u u_x`` with the 2nd-order central gradient, the 4th-order central Laplacian, periodic, stepped by
the explicit midpoint rule -- and that arithmetic is what ``libburgers_hip.so`` implements (pinned to the reference
class's own outputs in tests/test_burgers.py).  Everything around the step follows the Kuramoto-Sivashinsky env of the
same package: four Gaussian actuators (``GaussianForcing``), ``cfg_steps`` sub-steps per ``step`` with the forcing held,
l2-control reward ``-(1/N) |u|^2`` averaged over the sub-steps (taken before each update), truncation at
``Tmax``, fp32 observations ``[1, N]``.  **Parity unpinned** for all of that (no reference to compare with).

``BurgersBatchedVecEnv`` is the vector env (one HBM-resident fp32 batch, one launch per step, gym autoreset
semantics); ``BurgersEnv`` is the single-env view of it that ``gym.make`` returns.
"""
import math
from typing import Optional

import numpy as np
import torch

from pdegym._gym import gym
from pdegym.burgers import _hip
from pdegym.common.transforms import FuncTransform, GaussianForcing


def _stream(dev):
    return _hip.ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else 0))


def _ptr(t):
    return None if t is None else _hip.ctypes.c_void_p(t.data_ptr())


class BurgersBatchedVecEnv(gym.vector.VectorEnv):
    Xi = [0, 0.25, 0.5, 0.75]

    def __init__(self, num_envs: int, L: float = 2 * math.pi, N: int = 512, cfg_steps: int = 50, dt: float = 1e-3,
                 nu: float = 0.01, Tmax: float = 10.0, sigma: float = 0.15, ic_modes: int = 4, device: int = 0):
        self.L, self.N, self.cfg_steps, self.dt, self.nu, self.Tmax, self.sigma = L, N, cfg_steps, dt, nu, Tmax, sigma
        self.ic_modes = ic_modes
        self.dx = L / N
        self.x = np.linspace(0.0, L - L / N, N, dtype=np.float32)
        self.max_episode_steps = math.ceil(Tmax / (dt * cfg_steps))
        self.forcing = GaussianForcing(self.x, self.Xi, sigma, L, N)
        self.reward_func = FuncTransform(lambda obs, *a, **k: (-1.0) * (1 / self.N) * torch.norm(obs) ** 2)
        obs_space = gym.spaces.Box(-np.inf, np.inf, shape=(1, N), dtype=np.float32)
        act_space = gym.spaces.Box(-1.0, 1.0, shape=(1, len(self.Xi)), dtype=np.float32)
        super().__init__(num_envs, obs_space, act_space)
        self.device = torch.device("cuda", device)
        self._lib = _hip.load()
        self.u = torch.zeros((num_envs, N), dtype=torch.float32, device=self.device)
        self._F = self.forcing.forcing.to(self.device).contiguous()
        self._ssq = torch.zeros(num_envs, dtype=torch.float64, device=self.device)
        self._status = torch.zeros(num_envs, dtype=torch.int32, device=self.device)
        self.timestep = np.zeros(num_envs, dtype=np.int64)
        self._rngs = [np.random.RandomState() for _ in range(num_envs)]
        self._actions = None

    @property
    def scenario(self):
        return {"cfg_steps": self.cfg_steps, "L": self.L, "N": self.N, "dx": self.dx, "Tmax": self.Tmax, "dt": self.dt,
                "nu": self.nu, "Xi": self.Xi, "objective": "l2control"}

    def batched_reward_func(self, obs, phi=None):
        flat = obs.reshape(obs.shape[0], -1)
        if isinstance(obs, np.ndarray):
            return ((-1.0) * (1 / self.N) * np.linalg.norm(flat, axis=1) ** 2).astype(obs.dtype)
        return (-1.0) * (1 / self.N) * torch.linalg.vector_norm(flat, dim=1) ** 2

    # -- initial conditions: smooth random Fourier series (amplitude O(1)), per-env generator -----------------------
    def _initial(self, ids):
        rows = []
        for i in ids:
            rs = self._rngs[i]
            amp = rs.uniform(-1.0, 1.0, self.ic_modes) / np.arange(1, self.ic_modes + 1)
            ph = rs.uniform(0.0, 2 * np.pi, self.ic_modes)
            k = np.arange(1, self.ic_modes + 1)[:, None] * (2 * np.pi / self.L)
            rows.append((amp[:, None] * np.sin(k * self.x[None, :].astype(np.float64) + ph[:, None])).sum(0))
        return np.asarray(rows, dtype=np.float32)

    def _fresh_rows(self, ids):
        ids = np.asarray(ids, dtype=np.int64)
        u0 = torch.from_numpy(self._initial(ids)).to(self.device)
        self.u[torch.from_numpy(ids).to(self.device)] = u0
        self.timestep[ids] = 0
        return u0.cpu().numpy()

    # -- device-resident stepping -------------------------------------------------------------------------------
    def step_torch(self, actions: Optional[torch.Tensor], n_substeps: Optional[int] = None):
        """One env.step of the whole batch on torch's current stream: ``actions`` CUDA fp32 [E, 4] (or [E, 1, 4]) or
        None (no forcing).  Returns (state [E, N] fp32 -- the live buffer --, rewards [E] fp64); no host sync."""
        n = self.cfg_steps if n_substeps is None else int(n_substeps)
        a = None if actions is None else actions.reshape(self.num_envs, -1).contiguous()
        _hip.check(self._lib.bg_step(_stream(self.device), _ptr(self.u), _ptr(a), _ptr(self._F), len(self.Xi), self.num_envs,
                                     self.N, self.dx, self.dt, self.nu, n, None, _ptr(self._ssq), _ptr(self._status)))
        return self.u, self._ssq * (-(1.0 / self.N) / max(n, 1))

    def residual(self, u, phi=None):
        """nu u_xx - u u_x (+ phi) of fp32 rows [M, N] on the device (test hook of the kernel's stencils)."""
        u = torch.as_tensor(u, dtype=torch.float32, device=self.device).contiguous()
        phi_t = None if phi is None else torch.as_tensor(phi, dtype=torch.float32, device=self.device).contiguous()
        out = torch.empty_like(u)
        _hip.check(self._lib.bg_residual(_stream(self.device), _ptr(u), _ptr(phi_t), u.shape[0], u.shape[1], self.dx, self.nu,
                                         _ptr(out)))
        return out

    # -- gym.vector API ---------------------------------------------------------------------------------------------
    def reset_wait(self, seed=None, return_info: bool = False, options=None, **kwargs):
        if seed is None:
            seeds = [None] * self.num_envs
        elif isinstance(seed, (int, np.integer)):
            seeds = [int(seed) + i for i in range(self.num_envs)]
        else:
            seeds = list(seed)
        self._rngs = [np.random.RandomState(s) for s in seeds]
        obs = self._fresh_rows(np.arange(self.num_envs)).reshape(self.num_envs, 1, self.N)
        if return_info:
            return obs, {"step": self.timestep.copy()}
        return obs

    def reset(self, **kwargs):
        return self.reset_wait(**kwargs)

    def step_async(self, actions):
        self._actions = np.ascontiguousarray(np.asarray(actions, dtype=np.float32).reshape(self.num_envs, -1))

    def step_wait(self, **kwargs):
        assert self._actions is not None, "step_wait() without step_async()"
        a = torch.from_numpy(self._actions).to(self.device, non_blocking=True)
        self._actions = None
        u, rewards = self.step_torch(a)
        obs = u.cpu().numpy().reshape(self.num_envs, 1, self.N)
        rewards = rewards.cpu().numpy()
        if bool(self._status.any()):
            bad = np.nonzero(self._status.cpu().numpy())[0].tolist()
            raise FloatingPointError(f"non-finite Burgers state in envs {bad}")
        self.timestep += 1
        truncated = self.timestep >= self.max_episode_steps
        infos = {"step": self.timestep.copy()}
        if truncated.any():
            done = np.nonzero(truncated)[0]
            finals = np.full(self.num_envs, None, dtype=object)
            for i in done:
                finals[i] = obs[i].copy()
            infos["final_observation"], infos["_final_observation"] = finals, truncated.copy()
            obs[done] = self._fresh_rows(done).reshape(len(done), 1, self.N)
        return obs, rewards, np.zeros(self.num_envs, dtype=bool), truncated, infos


class BurgersEnv(gym.Env):
    """Single environment = a batch of one (same kernel, same arithmetic)."""

    def __init__(self, **config):
        super().__init__()
        self._config = dict(config)
        self._vec = None
        probe = {k: v for k, v in config.items() if k != "device"}
        L, N = probe.get("L", 2 * math.pi), probe.get("N", 512)
        self.L, self.N = L, N
        self.cfg_steps, self.dt = probe.get("cfg_steps", 50), probe.get("dt", 1e-3)
        self.max_episode_steps = math.ceil(probe.get("Tmax", 10.0) / (self.dt * self.cfg_steps))
        self.observation_space = gym.spaces.Box(-np.inf, np.inf, shape=(1, N), dtype=np.float32)
        self.action_space = gym.spaces.Box(-1.0, 1.0, shape=(1, 4), dtype=np.float32)

    @property
    def vec(self):
        if self._vec is None:   # constructing the env must not touch the GPU
            self._vec = BurgersBatchedVecEnv(1, **self._config)
        return self._vec

    def __getattr__(self, name):
        if name in ("forcing", "reward_func", "scenario", "dx", "x", "nu", "batched_reward_func"):
            return getattr(self.vec, name)
        raise AttributeError(name)

    def reset(self, seed=None, return_info=False, **kwargs):
        out = self.vec.reset(seed=seed, return_info=return_info)
        if return_info:
            return out[0][0], {"step": int(out[1]["step"][0])}
        return out[0]

    def step(self, action):
        self.vec.step_async(np.asarray(action, dtype=np.float32).reshape(1, -1))
        v = self.vec
        a = torch.from_numpy(v._actions).to(v.device)
        v._actions = None
        u, rewards = v.step_torch(a)
        if bool(v._status.any()):
            raise FloatingPointError("non-finite Burgers state")
        v.timestep += 1
        truncated = bool(v.timestep[0] >= v.max_episode_steps)
        return u.cpu().numpy().reshape(1, self.N), float(rewards[0]), False, truncated, {"step": int(v.timestep[0])}
