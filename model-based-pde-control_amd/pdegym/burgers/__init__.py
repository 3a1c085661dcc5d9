"""Burgers environments: the module the reference's ``pdegym/__init__.py:2`` imports and does not
ship.  Registers ``BurgersEnv-v0``; ``make_vec`` builds the batched HIP vector env."""
from pdegym._gym import gym
from pdegym.burgers.burgers import BurgersBatchedVecEnv, BurgersEnv

TimeLimit = gym.wrappers.TimeLimit
ENV_ID = "BurgersEnv-v0"


def make(config={}, new_step_api=True):
    env = BurgersEnv(**config)
    return TimeLimit(env, env.max_episode_steps, new_step_api=new_step_api)


def make_vec(num_envs, config={}, device=0):
    return BurgersBatchedVecEnv(num_envs, device=device, **config)


gym.envs.register(id=ENV_ID, entry_point="pdegym.burgers:make", order_enforce=False, new_step_api=True)
