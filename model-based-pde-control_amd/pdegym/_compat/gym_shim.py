"""Minimal stand-in for the slice of gym 0.25 that pdegym touches, used ONLY when the real
``gym`` package is not importable (it is absent from the build image and there is no network).

It is not installed into ``sys.modules``: pdegym reaches it through ``pdegym._gym.gym``.  With the
real gym present nothing in this file is used.  Semantics follow gym 0.25.2's documented
behaviour (new step API: 5-tuples; TimeLimit sets ``truncated``); the reference's own consumers of
these classes are pdegym/kuramoto/__init__.py:8-37 and pdegym/common/vec_wrappers.py.
"""
import types

import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None):
        self.shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)
        return [seed]


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
        super().__init__(shape, dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        bounded = np.isfinite(self.low) & np.isfinite(self.high)
        out = self._rng.normal(size=self.shape)
        out = np.where(bounded, self._rng.uniform(lo, hi, size=self.shape), out)
        return out.astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


def batch_space(space, n):
    return Box(np.broadcast_to(space.low, (n,) + space.shape), np.broadcast_to(space.high, (n,) + space.shape),
               shape=(n,) + space.shape, dtype=space.dtype)


class Env:
    metadata = {}
    reward_range = (-float("inf"), float("inf"))
    # (no class-level action_space / observation_space: wrappers forward them through __getattr__)

    def __init__(self, *args, **kwargs):
        pass

    @property
    def unwrapped(self):
        return self

    def close(self):
        pass


class Wrapper(Env):
    def __init__(self, env, new_step_api=True):
        self.env = env
        self.new_step_api = new_step_api

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)


class TimeLimit(Wrapper):
    """Sets ``truncated`` once ``max_episode_steps`` steps have elapsed since the last reset."""

    def __init__(self, env, max_episode_steps=None, new_step_api=True):
        super().__init__(env, new_step_api)
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = None

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            truncated = True
        return obs, reward, terminated, truncated, info

    def reset(self, **kwargs):
        self._elapsed_steps = 0
        return self.env.reset(**kwargs)


class VectorEnv(Env):
    def __init__(self, num_envs, observation_space, action_space, new_step_api=True):
        self.num_envs = num_envs
        self.is_vector_env = True
        self.single_observation_space = observation_space
        self.single_action_space = action_space
        self.observation_space = batch_space(observation_space, num_envs)
        self.action_space = batch_space(action_space, num_envs)
        self.closed = False
        self.new_step_api = new_step_api

    def reset_async(self, **kwargs):
        self._reset_kwargs = kwargs

    def reset_wait(self, **kwargs):
        raise NotImplementedError

    def reset(self, **kwargs):
        self.reset_async(**kwargs)
        return self.reset_wait(**kwargs)

    def step_async(self, actions):
        raise NotImplementedError

    def step_wait(self, **kwargs):
        raise NotImplementedError

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close_extras(self, **kwargs):
        pass

    def close(self, **kwargs):
        if not self.closed:
            self.close_extras(**kwargs)
            self.closed = True


class VectorEnvWrapper(VectorEnv):
    def __init__(self, env):
        assert isinstance(env, VectorEnv) or getattr(env, "is_vector_env", False)
        self.env = env

    def reset_async(self, **kwargs):
        return self.env.reset_async(**kwargs)

    def reset_wait(self, **kwargs):
        return self.env.reset_wait(**kwargs)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def step_async(self, actions):
        return self.env.step_async(actions)

    def step_wait(self):
        return self.env.step_wait()

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self, **kwargs):
        return self.env.close(**kwargs)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped


_REGISTRY = {}


def register(id, entry_point, **kwargs):
    _REGISTRY[id] = (entry_point, kwargs)


def _resolve(entry_point):
    if callable(entry_point):
        return entry_point
    import importlib
    mod, fn = entry_point.split(":")
    return getattr(importlib.import_module(mod), fn)


def make(id, **kwargs):
    if id not in _REGISTRY:
        raise KeyError(f"no registered env {id!r}")
    entry_point, spec_kwargs = _REGISTRY[id]
    call_kwargs = {k: v for k, v in spec_kwargs.items() if k not in ("order_enforce", "max_episode_steps")}
    call_kwargs.update(kwargs)
    return _resolve(entry_point)(**call_kwargs)


spaces = types.SimpleNamespace(Box=Box, Space=Space)
wrappers = types.SimpleNamespace(TimeLimit=TimeLimit)
envs = types.SimpleNamespace(register=register, registry=_REGISTRY)
vector = types.SimpleNamespace(VectorEnv=VectorEnv, VectorEnvWrapper=VectorEnvWrapper)
IS_SHIM = True
