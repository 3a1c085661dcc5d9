"""Single-env shape adapters for flat-vector agents (stable-baselines style), mirroring the
behaviour of the reference's ``pdegym/common/wrappers.py`` :5-30: the *spaces* lose the leading
singleton axis, actions get it back before reaching the env, and -- as in the reference -- the
observation is passed through ``np.expand_dims(obs, 0)``."""
import numpy as np

from pdegym._gym import gym


def _squeezed_box(space):
    low, high = np.squeeze(space.low, axis=0), np.squeeze(space.high, axis=0)
    return gym.spaces.Box(low, high, low.shape, dtype=np.float32)


class UnFlattenObsWrapper(gym.Wrapper):
    def __init__(self, env, new_step_api=False):
        super().__init__(env, new_step_api=new_step_api)
        self.observation_space = _squeezed_box(env.observation_space)

    def observation(self, obs):
        return np.expand_dims(obs, axis=0)

    def reset(self, **kwargs):
        out = self.env.reset(**kwargs)
        if isinstance(out, tuple):
            return (self.observation(out[0]),) + tuple(out[1:])
        return self.observation(out)

    def step(self, action):
        out = self.env.step(action)
        return (self.observation(out[0]),) + tuple(out[1:])


class UnFlattenActionWrapper(gym.Wrapper):
    def __init__(self, env, new_step_api=False):
        super().__init__(env, new_step_api=new_step_api)
        self.action_space = _squeezed_box(env.action_space)

    def action(self, action):
        return np.expand_dims(action, axis=0)

    def step(self, action):
        return self.env.step(self.action(action))
