"""KSBatchedVecEnv -- ``num_envs`` Kuramoto-Sivashinsky environments as ONE HBM-resident batch.

Replaces what ``gym.vector.make("KuramotoSivashinskyEnv-v0", num_envs=cpus)`` builds in the
reference (pdecontrol/mbrl/mbrl.py:81-86: an AsyncVectorEnv with one subprocess per env, each
running pdegym/kuramoto/kuramoto.py:78-116) with a single ``gym.vector.VectorEnv`` whose step is
one launch of the fused HIP stepper over all envs.  The vector-env contract the reference's
wrappers and ``Worker`` rely on (pdegym/common/vec_wrappers.py, pdecontrol/mbrl/worker.py:49-88)
is kept:

  * observations fp32 ``[E, 1, N]``, rewards fp64 ``[E]``, terminated / truncated bool ``[E]``
  * ``infos["step"]`` int array ``[E]``
  * autoreset: when an env truncates, ``infos["final_observation"][i]`` holds its last
    observation (``None`` elsewhere), ``infos["_final_observation"]`` is the mask, and the returned
    observation is the one after reset (fresh IC + 200 000 sub-step burn-in, run on the GPU for
    just those envs; in ``reset_mode`` arithmetic -- "fast" by default, "exact" = the reference's
    realisation bit for bit, see kuramoto.py).
  * seeding: ``reset(seed=s)`` seeds env i with ``s + i`` (gym's vector convention); the IC of env
    i is then exactly what the reference's ``reset(seed=s+i)`` draws (``mt_batch.BatchedMT19937``: the
    per-env MT19937 streams of all envs as array operations, bit-identical to one ``RandomState`` per env).

Multi-GPU: envs are independent, so a job shards them by rank with ``shard_envs`` and every rank
builds its own KSBatchedVecEnv on its own GPU; no collective is involved.  A SINGLE controller process (what
pdecontrol/mbrl/script.py is) reaches several GPUs through ``KSShardedVecEnv`` (pdegym/kuramoto/sharded.py).
"""
from typing import Optional, Sequence

import numpy as np

from pdegym._gym import gym
from pdegym.kuramoto.kuramoto import KuramotoSivashinskyEnv, default_reset_mode
from pdegym.kuramoto.mt_batch import BatchedMT19937

DEFAULT_RESET_MODE = default_reset_mode


def shard_envs(num_envs: int, rank: int, world_size: int):
    """Contiguous block of env ids owned by ``rank``: [lo, hi)."""
    per, extra = divmod(num_envs, world_size)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


class KSBatchedVecEnv(gym.vector.VectorEnv):
    def __init__(self, num_envs: int, config: Optional[dict] = None, device: int = 0, step_mode: str = "fast",
                 reset_mode: Optional[str] = None, variant: str = "auto", burn_in: bool = True, _stepper_cls=None):
        config = dict(config or {})
        # an env config may carry the stepper options too (that is how gym.make(id, config=...) passes them)
        step_mode = config.pop("step_mode", step_mode)
        reset_mode = config.pop("reset_mode", reset_mode) or default_reset_mode()
        device = config.pop("device", device)
        variant = config.pop("variant", variant)
        # a (never stepped) single env supplies spaces, forcing matrix, reward function, constants
        self.proto = KuramotoSivashinskyEnv(**config)
        p = self.proto
        if not p.objective:
            raise NotImplementedError("the batched env implements the l2control reward only")
        super().__init__(num_envs, p.observation_space, p.action_space)
        self.N, self.L, self.dt, self.cfg_steps = p.N, p.L, p.dt, p.cfg_steps
        self.max_episode_steps = p.max_episode_steps
        self.burn_in_substeps = int(p.BURN_IN_TIME / p.dt / p.cfg_steps) * p.cfg_steps if burn_in else 0
        self.device, self.step_mode, self.reset_mode = device, step_mode, reset_mode
        if _stepper_cls is None:
            import kspde  # fails loudly if libkspde.so is missing
            _stepper_cls = kspde.KSStepper
        self._build_steppers(_stepper_cls, variant)
        self.timestep = np.zeros(num_envs, dtype=np.int64)
        self._rng = BatchedMT19937(num_envs)      # one MT19937 stream per env (unseeded until reset(seed=...))
        self._actions = None

    def _build_steppers(self, stepper_cls, variant):
        self.stepper = stepper_cls(self.num_envs, self.N, self.L, self.dt, device=self.device, mode=self.step_mode,
                                   variant=variant)
        self.stepper.set_forcing(self.proto.forcing.forcing.numpy())

    # attributes the reference reads through ``env.unwrapped`` / ``get_attr``
    @property
    def forcing(self):
        return self.proto.forcing

    @property
    def reward_func(self):
        return self.proto.reward_func

    @property
    def scenario(self):
        return self.proto.scenario

    # -- helpers -----------------------------------------------------------------------------
    def _set_mode(self, mode):
        if self.stepper.mode != mode:
            self.stepper.set_mode(mode)

    def _raise_on(self, status):
        if np.any(status):
            raise FloatingPointError(f"overflow encountered in envs {np.nonzero(status)[0].tolist()}")

    def _fresh_rows(self, ids: Sequence[int]):
        """New ICs for ``ids`` (kuramoto.py:106) and their burn-in (:108-109) on the GPU."""
        ids = np.asarray(ids, dtype=np.int32)
        u0 = self._rng.uniform_rows(ids, -0.4, 0.4, self.N)
        self.stepper.set_state_rows(ids, u0)
        self._set_mode(self.reset_mode)
        obs, _, status = self.stepper.step_rows(ids, self.burn_in_substeps)
        self._set_mode(self.step_mode)
        self._raise_on(status)
        self.timestep[ids] = 0
        return obs

    # -- gym.vector API ----------------------------------------------------------------------
    def reset_wait(self, seed=None, return_info: bool = False, options=None, **kwargs):
        if seed is None:
            seeds = [None] * self.num_envs
        elif isinstance(seed, (int, np.integer)):
            seeds = [int(seed) + i for i in range(self.num_envs)]
        else:
            seeds = list(seed)
            assert len(seeds) == self.num_envs
        self._rng.seed_rows(np.arange(self.num_envs), seeds)
        obs = self._fresh_rows(np.arange(self.num_envs))
        obs = obs.reshape(self.num_envs, 1, self.N)
        if return_info:
            return obs, {"step": self.timestep.copy()}
        return obs

    def reset(self, **kwargs):
        return self.reset_wait(**kwargs)

    def step_async(self, actions):
        a = np.asarray(actions, dtype=np.float32).reshape(self.num_envs, -1)
        self._actions = np.ascontiguousarray(a)

    def step_wait(self, **kwargs):
        assert self._actions is not None, "step_wait() without step_async()"
        obs, ssq, status = self.stepper.step_actions(self._actions, self.cfg_steps)
        self._actions = None
        return self._finish_step(obs, ssq, status)

    def _finish_step(self, obs, ssq, status):
        """Rewards, episode bookkeeping and autoreset from the raw outputs of one step of every env."""
        self._raise_on(status)
        rewards = (-1.0) * (1 / self.N) * ssq / self.cfg_steps
        self.timestep += 1
        truncated = self.timestep >= self.max_episode_steps
        terminated = np.zeros(self.num_envs, dtype=bool)
        infos = {"step": self.timestep.copy(), "_step": np.ones(self.num_envs, dtype=bool)}
        obs = obs.reshape(self.num_envs, 1, self.N)
        if truncated.any():
            done = np.nonzero(truncated)[0]
            finals = np.full(self.num_envs, None, dtype=object)
            for i in done:
                finals[i] = obs[i].copy()
            infos["final_observation"] = finals
            infos["_final_observation"] = truncated.copy()
            obs[done] = self._fresh_rows(done).reshape(len(done), 1, self.N)
        return obs, rewards, terminated, truncated, infos

    # -- device-resident stepping (no PCIe on the hot path) -----------------------------------
    def step_torch(self, actions):
        """``step`` for agents that live on the same GPU: ``actions`` is a CUDA fp32 tensor ``[E, 1, 4]``
        (or ``[E, 4]``); returns ``(obs [E,1,N] fp32, rewards [E] fp64, truncated [E] bool)`` as CUDA
        tensors written by the kernel, asynchronously on torch's current stream.  Episode bookkeeping
        (timestep / truncation) stays on the host; autoreset is the caller's job (``reset_rows``)."""
        import torch
        assert actions.is_cuda and actions.dtype == torch.float32
        a = actions.reshape(self.num_envs, -1).contiguous()
        dev = a.device
        if getattr(self, "_dev_out", None) is None or self._dev_out[0].device != dev:
            self._dev_out = (torch.empty((self.num_envs, 1, self.N), dtype=torch.float32, device=dev),
                             torch.empty(self.num_envs, dtype=torch.float64, device=dev),
                             torch.zeros(self.num_envs, dtype=torch.int32, device=dev))
        stream = torch.cuda.current_stream(dev).cuda_stream
        if getattr(self, "_stream_handle", None) != stream:
            self.stepper.set_stream(stream)
            self._stream_handle = stream
        obs, ssq, status = self._dev_out
        self.stepper.step_device(d_actions=a.data_ptr(), n_substeps=self.cfg_steps, d_obs=obs.data_ptr(),
                                 d_ssq=ssq.data_ptr(), d_status=status.data_ptr())
        rewards = ssq * (-(1.0 / self.N) / self.cfg_steps)
        self.timestep += 1
        truncated = torch.from_numpy(self.timestep >= self.max_episode_steps).to(dev, non_blocking=True)
        return obs, rewards, truncated, status

    def reset_rows(self, ids):
        """Fresh IC + burn-in for the listed envs (the autoreset of ``step_wait`` as an explicit call)."""
        return self._fresh_rows(np.asarray(ids, dtype=np.int32))

    def close_extras(self, **kwargs):
        self.stepper.close()
