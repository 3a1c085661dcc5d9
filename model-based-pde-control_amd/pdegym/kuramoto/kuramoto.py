"""Kuramoto-Sivashinsky gym environment backed by the MI355X HIP stepper.

Drop-in for the reference's ``pdegym/kuramoto/kuramoto.py`` (class name, module path, constructor
keywords and defaults :29-41, attributes, ``step`` :78-98, ``reset`` :100-116, ``rhs`` :118-129,
``time`` / ``scenario`` :131-150).  Nothing numerical happens in this file: the RK4 / finite
difference work is one launch of ``libkspde.so`` per ``step`` (all ``cfg_steps`` sub-steps fused)
and one launch per ``reset`` (the whole 200 000 sub-step burn-in).  Without the HIP library or a GPU, stepping
raises; ``device=-1`` (``"cpu"``) asks for the library's CPU twin explicitly (BASELINE configs[0]).

Differences from the reference that a caller can observe:
  * ``env.u`` is a property backed by device memory (reads copy D2H, assignment copies H2D).
  * Arithmetic modes (``step_mode`` / ``reset_mode``): "fast" (merged stencil, FMAs: <= 1e-12 from the reference per
    sub-step; the contract is 1e-9 per sub-step) is the default of BOTH since round 3; "exact" keeps the reference's
    operation order and is bit-identical to it -- including a seeded ``reset()``, whose 200 000 chaotic sub-steps
    amplify fast mode's 1e-16 rounding differences to O(1e-3), i.e. a different but statistically equivalent
    realisation of the same attractor.  The burn-in is two thirds of all sub-steps of a run and exact arithmetic costs
    2.6x (2.13 vs 0.83 ms per 250 sub-steps at 4096 x 256), so the default follows the per-sub-step contract;
    ``reset_mode="exact"`` or ``PDEGYM_RESET_MODE=exact`` is the parity switch (every parity test passes it).
  * ``objective=""`` (the ``dissipation`` reward) raises in ``step`` here with NotImplementedError;
    in the reference it raises TypeError (FuncTransform hands tensors to scipy), so no working
    behaviour is lost.  ``reward_func`` itself is provided for both objectives.
"""
import math
import os
from typing import Dict, List

import numpy as np
import torch

from pdegym._gym import gym
from pdegym.common.transforms import FuncTransform, GaussianForcing


def default_reset_mode() -> str:
    """Arithmetic of the reset burn-in when the constructor is not told: ``PDEGYM_RESET_MODE`` or "fast"."""
    mode = os.environ.get("PDEGYM_RESET_MODE", "fast").strip().lower()
    if mode not in ("fast", "exact"):
        raise ValueError(f"PDEGYM_RESET_MODE must be 'fast' or 'exact', not {mode!r}")
    return mode


class KuramotoSivashinskyEnv(gym.Env):
    metadata = {"render.modes": ["rgb_array"]}

    Xi = [0, 0.25, 0.5, 0.75]  # relative actuator positions
    eps = np.finfo(np.float32).eps
    reward_range = (-float("inf"), float("inf"))

    #: sub-steps of the reset burn-in in physical time (reference: int(200.0 / dt / cfg_steps) steps)
    BURN_IN_TIME = 200.0

    def __init__(
        self,
        L: float = 22.0,
        N: int = 64,
        cfg_steps: int = 250,
        Ttrans: int = 40,
        Tmax: float = 100.0,
        dt=0.001,
        noise: float = 0.1,
        sigma: float = 0.4,
        lmbda: float = 0.0,
        objective: str = "dissipation",
        device: int = 0,
        step_mode: str = "fast",
        reset_mode: str = None,
        variant: str = "auto",
        _stepper_cls=None,
    ):
        super().__init__()
        reset_mode = reset_mode or default_reset_mode()
        self.L, self.N, self.cfg_steps = L, N, cfg_steps
        self.Ttrans, self.Tmax, self.dt = Ttrans, Tmax, dt
        self.noise, self.sigma, self.lmbda, self.objective = noise, sigma, lmbda, objective

        self.dx = self.L / self.N
        self.x = np.linspace(0.0, self.L - self.L / self.N, self.N, dtype=np.float32)
        self.max_episode_steps = math.ceil(self.Tmax / (self.dt * self.cfg_steps))

        self.forcing = GaussianForcing(self.x, self.Xi, self.sigma, self.L, self.N)
        self.noop = np.zeros((1, len(self.Xi)), dtype=np.float32)

        # NB: like the reference (kuramoto.py:72) the *truthiness* of ``objective`` selects the
        # reward, so the default string "dissipation" selects l2control.
        self.reward_func = FuncTransform(self._l2control if self.objective else self._dissipation)

        self.action_space = gym.spaces.Box(-1.0, 1.0, shape=(1, len(self.Xi)), dtype=np.float32)
        self.observation_space = gym.spaces.Box(-np.inf, np.inf, shape=(1, self.N), dtype=np.float32)

        if isinstance(device, str):
            device = -1 if device.lower() == "cpu" else int(device)
        self.device, self.step_mode, self.reset_mode, self.variant = device, step_mode, reset_mode, variant
        self._stepper_cls = _stepper_cls
        self._stepper = None  # created lazily: constructing the env must not touch the GPU
        self._pending_u = None
        self.timestep = 0

    # -- rewards -----------------------------------------------------------------------------
    def _l2control(self, obs, *args, **kwargs):
        return (-1.0) * (1 / self.N) * torch.norm(obs) ** 2

    def batched_reward_func(self, obs, phi=None):
        """``reward_func`` for a whole batch ``[B, ..., N]`` at once (numpy or torch, host or device): the l2control
        value of every row, in the input's dtype -- what the reference evaluates sample by sample in a Python loop
        (pdecontrol/mbrl/world/world.py:170).  Only the l2control objective has a batched form."""
        if not self.objective:
            raise NotImplementedError("batched reward exists for the l2control objective only")
        if isinstance(obs, np.ndarray):
            flat = obs.reshape(obs.shape[0], -1)
            return ((-1.0) * (1 / self.N) * np.linalg.norm(flat, axis=1) ** 2).astype(obs.dtype)
        flat = obs.reshape(obs.shape[0], -1)
        return (-1.0) * (1 / self.N) * torch.linalg.vector_norm(flat, dim=1) ** 2

    def _dissipation(self, obs, phi, *args, **kwargs):
        as_np = lambda v: np.squeeze(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
        u, p = as_np(obs).astype(np.float64), as_np(phi)
        _, (u_x, u_xx, _) = self.rhs(u, p)
        return torch.as_tensor((-1) * ((u_xx * u_xx).mean() + (u_x * u_x).mean() + (u * p).mean()))

    # -- device state ------------------------------------------------------------------------
    @property
    def stepper(self):
        if self._stepper is None:
            if self._stepper_cls is None:
                import kspde  # fails loudly if libkspde.so is missing
                self._stepper_cls = kspde.KSStepper
            self._stepper = self._stepper_cls(1, self.N, self.L, self.dt, device=self.device, mode=self.step_mode,
                                              variant=self.variant)
            self._stepper.set_forcing(self.forcing.forcing.numpy())
            if self._pending_u is not None:
                self._stepper.set_state(self._pending_u[None, :])
                self._pending_u = None
        return self._stepper

    @property
    def u(self):
        if self._stepper is None and self._pending_u is not None:
            return self._pending_u
        return self.stepper.get_state()[0]

    @u.setter
    def u(self, value):
        value = np.ascontiguousarray(np.asarray(value, dtype=np.float64).reshape(self.N))
        if self._stepper is None:
            self._pending_u = value
        else:
            self._stepper.set_state(value[None, :])

    def _advance(self, action, n_substeps, mode):
        s = self.stepper
        if s.mode != mode:
            s.set_mode(mode)
        if action is None:
            _, ssq, status = s.step(None, n_substeps, want_obs=False)
        else:
            _, ssq, status = s.step_actions(action, n_substeps, want_obs=False)
        if status[0]:
            # the reference traps this with np.seterr(over="raise") (kuramoto.py:12)
            raise FloatingPointError("overflow encountered in Kuramoto-Sivashinsky state")
        return ssq[0]

    # -- gym API -----------------------------------------------------------------------------
    def step(self, action: List):
        if not self.objective:
            raise NotImplementedError("the 'dissipation' reward is not available inside step() "
                                      "(the reference raises TypeError on this path)")
        action = np.array(action, dtype=np.float32)
        # phi = forcing(action) is evaluated inside the kernel as the fp32 FMA chain
        # a0*F0 (+) a1*F1 (+) a2*F2 (+) a3*F3 -- what torch's CPU matmul produced where the golden
        # vectors were captured; a host-side matmul is not bit-stable across CPU microarchitectures.
        ssq = self._advance(action.reshape(1, -1), self.cfg_steps, self.step_mode)
        reward = (-1.0) * (1 / self.N) * ssq / self.cfg_steps

        self.timestep += 1
        truncated = self.timestep >= self.max_episode_steps
        obs = self.u.reshape(1, -1)
        return obs, reward, False, truncated, {"step": self.timestep}

    def reset(self, seed: int = None, return_info=False, **kwargs) -> Dict:
        np.random.seed(seed)  # the reference seeds numpy's global generator (kuramoto.py:101)
        tsteps = int(self.BURN_IN_TIME / self.dt / self.cfg_steps)
        self.u = np.random.uniform(-0.4, 0.4, size=self.N)
        # tsteps no-op steps == tsteps * cfg_steps sub-steps with phi = 0, fused into one launch
        self._advance(None, tsteps * self.cfg_steps, self.reset_mode)
        self.timestep = 0
        obs = self.u.reshape(1, -1)
        if return_info:
            return obs, {"step": self.timestep}
        return obs

    def rhs(self, u, phi):
        u_arr = np.asarray(u, dtype=np.float64)
        shape = u_arr.shape
        phi_arr = np.broadcast_to(np.asarray(phi, dtype=np.float32), shape)
        outs = self.stepper.rhs(u_arr.reshape(-1, self.N), phi_arr.reshape(-1, self.N))
        rhs, u_x, u_xx, u_xxxx = (o.reshape(shape) for o in outs)
        return rhs, (u_x, u_xx, u_xxxx)

    def close(self):
        if self._stepper is not None:
            self._stepper.close()
            self._stepper = None

    @property
    def time(self):
        return self.timestep * self.cfg_steps * self.dt

    @property
    def scenario(self):
        # the reference reports constants for noise / lmbda here (kuramoto.py:146-147)
        return {
            "cfg_steps": self.cfg_steps, "Ttrans": self.Ttrans, "L": self.L, "N": self.N, "dx": self.dx,
            "Tmax": self.Tmax, "dt": self.dt, "Xi": self.Xi, "noise": 0.1, "lmbda": 1.0,
            "objective": self.objective,
        }
