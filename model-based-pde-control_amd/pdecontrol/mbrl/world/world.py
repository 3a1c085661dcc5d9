"""Surrogate-backed vector environment for imagined rollouts.

API / behaviour mirror of the reference's ``pdecontrol/mbrl/world/world.py`` (``BaseWorldVecEnv``
:15-65, ``WorldVecEnv`` :68-204): ``reset`` samples warm-up windows from the replay and rolls the
surrogate (ensemble) over them, ``step_async`` advances every imagined trajectory by one surrogate
step under ``no_grad`` and evaluates the reward, ``step_wait`` truncates ALL trajectories together at
the rollout horizon / env time limit and restarts them.

Device handling (the reference is CPU-only): tensors follow the surrogate's parameters -- the warm-up
batch and the actions are moved to that device, observations / rewards come back as numpy at the gym
boundary.  ``batched_reward_func`` (optional) evaluates the reward for the whole batch at once; the
reference's per-sample Python loop (``world.py:170``) remains the default so arbitrary reward callables
keep working.
"""
from typing import Any, Callable, Sequence

import numpy as np
import torch
from torch.utils.data import Dataset, RandomSampler

from pdegym._gym import gym
from pdecontrol.mbrl.types import ModelRollout
from pdecontrol.surrogates.common.dataset import PDEDataLoader

try:  # real gym
    from gym.vector.utils.spaces import batch_space  # type: ignore
except ImportError:
    from pdegym._compat.gym_shim import batch_space


def _surrogate_device(surrogate):
    modules = getattr(surrogate, "modules", None)
    if isinstance(modules, (list, tuple)) and modules:     # PDEEnsemble keeps a plain list of training modules
        return next(modules[0].surrogate.parameters()).device
    return next(surrogate.parameters()).device


class BaseWorldVecEnv(gym.vector.VectorEnv):
    def __init__(self, surrogate, observation_space, action_space, max_episode_steps: int, stransf,
                 reward_func: Callable, num_envs: int, horizon: int, tstep: float):
        self.surrogate, self.max_episode_steps, self.stransf = surrogate, max_episode_steps, stransf
        self.reward_func, self.num_envs, self.horizon, self.tstep = reward_func, num_envs, horizon, tstep
        self.observation_space = gym.spaces.Box(observation_space.low, observation_space.high)
        self.action_space = gym.spaces.Box(action_space.low, action_space.high)
        self.is_vector_env = True

    def setup(self, starting: Dataset):
        """Endless stream of warm-up batches, sampled with replacement from ``starting``."""
        sampler = RandomSampler(starting, replacement=True, num_samples=int(1e10))
        self.loader = iter(PDEDataLoader(starting, batch_size=self.num_envs, shuffle=False, sampler=sampler,
                                         drop_last=True, collate_fn=PDEDataLoader.padding_collate))


class WorldVecEnv(BaseWorldVecEnv):
    def __init__(self, surrogate, observation_space, action_space, max_episode_steps: int, stransf,
                 reward_func: Callable, num_envs: int, horizon: int, tstep: float, batched_reward_func=None):
        super().__init__(surrogate, observation_space, action_space, max_episode_steps, stransf, reward_func, num_envs,
                         horizon, tstep)
        self.batched_reward_func = batched_reward_func
        # spaces as seen through the replay->world transforms (stransf is their inverse)
        unbatch = lambda fn, x: np.squeeze(fn(x[np.newaxis, ...]), axis=0)
        low = unbatch(self.stransf.atransf.Inverse, action_space.low)
        high = unbatch(self.stransf.atransf.Inverse, action_space.high)
        self.single_action_space = gym.spaces.Box(low, high, shape=low.shape)
        self.action_space = batch_space(self.single_action_space, self.num_envs)
        obs = unbatch(self.stransf.otransf.Inverse, observation_space.sample())
        self.single_observation_space = gym.spaces.Box(-np.inf, np.inf, shape=obs.shape)
        self.observation_space = batch_space(self.single_observation_space, self.num_envs)
        self.tmp = None

    def _host_obs(self):
        return self.output.outputs.detach().squeeze(1).cpu().numpy()

    def reset(self, **kwargs):
        self.surrogate.eval()
        with torch.no_grad():
            states, actions, _, _, _, _, steps = next(self.loader)
            dev = _surrogate_device(self.surrogate)
            times = self.tstep * torch.arange(actions.size(1))
            targets = self.tstep * actions.size(1)
            self.output: ModelRollout = self.surrogate.rollout(states=states.to(dev), actions=actions.to(dev),
                                                                hidden=None, times=times, targets=targets)
        self.timesteps = steps[:, -1].numpy()  # env step counter after the warm-up window
        self.simulated = 0
        self.tmp = None
        self.surrogate.train()
        obs = self._host_obs()
        if kwargs.get("return_info", False):
            return obs, {"step": self.timesteps.copy()}
        return obs

    def step_async(self, actions: Sequence[Any]) -> None:
        self.surrogate.eval()
        self.simulated += 1
        self.timesteps += 1
        with torch.no_grad():
            host_actions = np.array(actions, dtype=np.float32)
            dev = self.output.outputs.device
            # the surrogate expects [B, T, C, A]; the env interface passes [B, C, A]
            act = torch.from_numpy(host_actions).to(dev).unsqueeze(1)
            self.output = self.surrogate.rollout(states=self.output.outputs, actions=act, hidden=self.output.hidden,
                                                 times=0.0, targets=self.tstep)
            obs = self._host_obs()
            orescaled = self.stransf.otransf(obs)          # back to the env's observation scale
            arescaled = self.stransf.atransf(host_actions)
            if self.batched_reward_func is not None:
                rewards = np.asarray(self.batched_reward_func(orescaled, arescaled), dtype=np.float32)
            else:
                rewards = np.asarray([self.reward_func(o, a) for o, a in zip(orescaled, arescaled)], dtype=np.float32)
        self.tmp = rewards
        self.surrogate.train()

    def step_wait(self, **kwargs: Any):
        obs = self.output.outputs.detach().cpu().numpy().squeeze(1)
        rewards = self.tmp
        env_limit = np.broadcast_to(self.timesteps >= self.max_episode_steps, (self.num_envs,))
        rll_limit = np.broadcast_to(self.simulated >= self.horizon, (self.num_envs,))
        # all trajectories are cut together, once every one of them has hit a limit
        truncated = np.broadcast_to(np.all(env_limit | rll_limit), (self.num_envs,))
        terminated = np.zeros(self.num_envs, dtype=np.bool_)
        infos = {"step": self.timesteps.copy()}
        if np.any(truncated):
            infos["_final_observation"] = truncated.copy()
            infos["final_observation"] = obs[truncated]
            obs = self.reset()
        return obs, rewards, terminated, truncated, infos
