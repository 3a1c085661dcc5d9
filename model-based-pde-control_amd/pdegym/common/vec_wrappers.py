"""Vector-env wrappers of the experience-collection stack.

API / behaviour mirror of the reference's ``pdegym/common/vec_wrappers.py``
(``StoreNObsVecWrapper`` :10-52, ``StoreNActionsVecWrapper`` :55-95, ``TransformActionWrapper`` :98-129,
``TransformObsWrapper`` :132-190) as stacked by ``pdecontrol/mbrl/mbrl.py:257-291``.  They are pure
host-side bookkeeping on ``[E, ...]`` numpy batches and run unchanged on top of the HBM-resident
``KSBatchedVecEnv`` (one array per step for all E envs, no per-env Python loops here; the transforms
they call are batch-vectorised, see pdegym/common/transforms.py::BatchTransform).

Pinned against the reference's own classes: oracle/gen_golden.py drives the reference wrappers and
these through the same scripted fake vector env (tests/_fake_vec_env.py); tests/test_vec_wrappers.py
compares every array bit for bit.
"""
from copy import deepcopy
from typing import Any, Sequence

import numpy as np

from pdegym._gym import gym

_Wrapper = gym.vector.VectorEnvWrapper


def _reset_through(env, kwargs):
    """env.reset(**kwargs) -> (obs, info or None) whichever form ``return_info`` selects."""
    if kwargs.get("return_info", False):
        obs, info = env.reset(**kwargs)
        return obs, info
    return env.reset(**kwargs), None


def _reset_result(obs, info, kwargs):
    return (obs, info) if kwargs.get("return_info", False) else obs


def _final_rows(infos):
    """Final observations of the envs that just finished, as one fp32 array (rows in env order).
    The reference converts the whole ``infos["final_observation"]`` object array, which only works
    when EVERY env finished on this step; entries that are None (partial autoreset) are skipped here."""
    return np.asarray([f for f in infos["final_observation"] if f is not None], dtype=np.float32)


class _History:
    """Last ``num_steps`` items per env, newest in the last slot, with a validity mask."""

    def __init__(self, num_envs, num_steps, item_shape, dtype):
        self.values = np.zeros((num_envs, num_steps) + tuple(item_shape), dtype=dtype)
        self.mask = np.zeros((num_envs, num_steps), dtype=np.bool_)


def _push(history, items):
    """The reference writes the new item into slot 0 and rotates left with ``np.roll`` (a fresh copy of the whole
    history per step).  Same resulting array, shifted in place: older items move one slot towards 0, the new item
    lands in the last slot.  (The controller's stacks use ``num_steps = 1``, where this is a plain overwrite; the
    reference's consumers copy what they read, pdecontrol/mbrl/worker.py:50,70,75.)"""
    if history.shape[1] > 1:
        history[:, :-1] = history[:, 1:]
    history[:, -1] = items


class StoreNObsVecWrapper(_Wrapper):
    """Keeps the last ``num_steps`` observations (``.obs``, ``.mask``) and the final observations
    of finished episodes (``.finals``) for the world-model wrapper further up the stack."""

    def __init__(self, env, num_steps: int = 1) -> None:
        super().__init__(env)
        self.num_steps = num_steps
        space = self.observation_space
        hist = _History(self.env.num_envs, num_steps, space.shape[1:], space.dtype)
        self.obs, self.mask = hist.values, hist.mask
        self.finals = np.zeros_like(hist.values)

    def step_wait(self, **kwargs: Any):
        obs, rewards, terminated, truncated, infos = self.env.step_wait()
        if "final_observation" in infos:
            done = infos["_final_observation"]
            self.finals[done] = np.expand_dims(_final_rows(infos), axis=1)
            self.mask[done] = False
        _push(self.obs, obs)
        _push(self.mask, True)
        return obs, rewards, terminated, truncated, infos

    def reset(self, **kwargs) -> Any:
        obs, info = _reset_through(self.env, kwargs)
        self.obs = np.repeat(obs[:, np.newaxis, ...], self.num_steps, axis=1)
        self.mask[:, :-1] = False
        self.mask[:, -1] = True
        return _reset_result(obs, info, kwargs)


class StoreNActionsVecWrapper(_Wrapper):
    """Keeps the last ``num_steps`` actions (``.actions``, ``.mask``)."""

    def __init__(self, env, num_steps: int = 1) -> None:
        super().__init__(env)
        self.num_steps = num_steps
        space = self.action_space
        # (sic) the reference allocates the action history with the OBSERVATION dtype
        hist = _History(self.env.num_envs, num_steps, space.shape[1:], self.observation_space.dtype)
        self.actions, self.mask = hist.values, hist.mask

    def step_async(self, actions: Sequence[Any]) -> None:
        _push(self.actions, actions)
        _push(self.mask, True)
        return super().step_async(actions)

    def step_wait(self, **kwargs: Any):
        obs, rewards, terminated, truncated, infos = self.env.step_wait()
        if "final_observation" in infos:
            self.mask[infos["_final_observation"], :-1] = False
        return obs, rewards, terminated, truncated, infos

    def reset(self, **kwargs) -> Any:
        obs, info = _reset_through(self.env, kwargs)
        self.mask[:, :] = False
        return _reset_result(obs, info, kwargs)


class TransformActionWrapper(_Wrapper):
    """Agent-side actions -> env-side actions through ``transform`` (statistics updated unless frozen)."""

    def __init__(self, env, transform, frozen=False):
        super().__init__(env)
        self.transform, self.frozen = transform, frozen
        low, high = self.transform.Inverse(self.env.action_space.low), self.transform.Inverse(self.env.action_space.high)
        self.action_space = gym.spaces.Box(low, high, shape=low.shape)
        self.single_action_space = gym.spaces.Box(low[0], high[0], shape=low.shape[1:])

    def step_async(self, actions: Sequence[Any]) -> None:
        if not self.frozen:
            self.transform.update(actions)
        return self.env.step_async(self.transform(actions))

    def step_wait(self, **kwargs: Any):
        return self.env.step_wait(**kwargs)

    def reset(self, **kwargs) -> Any:
        return self.env.reset(**kwargs)


class TransformObsWrapper(_Wrapper):
    """Env-side observations -> agent/world-side observations through ``transform``."""

    def __init__(self, env, transform, frozen=False):
        super().__init__(env)
        self.transform, self.frozen = transform, frozen
        clean = lambda a: np.nan_to_num(a, nan=-np.inf, posinf=np.inf, neginf=-np.inf)
        low = clean(self.transform(self.env.observation_space.low))
        high = clean(self.transform(self.env.observation_space.high))
        self.observation_space = gym.spaces.Box(low, high, shape=low.shape)
        self.single_observation_space = gym.spaces.Box(low[0], high[0], shape=low.shape[1:])

    def _apply(self, obs):
        if not self.frozen:
            self.transform.update(obs)
        return self.transform(obs)

    def step_wait(self, **kwargs: Any):
        obs, rewards, terminated, truncated, infos = self.env.step_wait()
        obs = self._apply(obs)
        if "final_observation" in infos:
            finals = _final_rows(infos)
            # (sic) as in the reference (vec_wrappers.py:160-166) the final observations are only
            # transformed -- and written back -- when the wrapper is NOT frozen
            if not self.frozen:
                self.transform.update(finals)
                finals = self.transform(finals)
                if len(finals) == len(infos["final_observation"]):
                    infos["final_observation"] = finals          # every env finished: plain array, as the reference
                else:                                             # partial autoreset: keep None for running envs
                    out = np.full(len(infos["final_observation"]), None, dtype=object)
                    for i, row in zip(np.nonzero(infos["_final_observation"])[0], finals):
                        out[i] = row
                    infos["final_observation"] = out
        return obs, rewards, terminated, truncated, infos

    def reset(self, **kwargs) -> Any:
        obs, info = _reset_through(self.env, kwargs)
        if info is not None:
            self.info = deepcopy(info)
        return _reset_result(self._apply(obs), info, kwargs)
