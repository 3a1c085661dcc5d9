"""Factories + gym registration of the Kuramoto-Sivashinsky environments.

Mirror of the reference's ``pdegym/kuramoto/__init__.py`` (``make`` :8-12, ``make_sb3`` :15-23,
registration of ``KuramotoSivashinskyEnv-v0`` / ``KuramotoSivashinskyEnvSB3-v0`` :26-37), plus
``make_vec``: the HBM-resident batched vector env that replaces ``gym.vector.make(id, num_envs)``.
"""
import os

from pdegym._gym import gym
from pdegym.kuramoto.kuramoto import KuramotoSivashinskyEnv
from pdegym.kuramoto.batched import KSBatchedVecEnv, shard_envs
from pdegym.kuramoto.sharded import KSShardedVecEnv, resolve_devices

TimeLimit = gym.wrappers.TimeLimit

ENV_ID = "KuramotoSivashinskyEnv-v0"
ENV_ID_SB3 = "KuramotoSivashinskyEnvSB3-v0"


def _with_device(config):
    """``PDEGYM_DEVICE`` (``cpu`` or a HIP ordinal) chooses the stepper's device for envs whose config does not: the
    route by which an unmodified caller runs BASELINE configs[0] on the library's CPU twin."""
    dev = os.environ.get("PDEGYM_DEVICE", "")
    if dev and "device" not in config:
        config = dict(config, device=-1 if dev.lower() == "cpu" else int(dev))
    return config


def make(config={}, new_step_api=True):
    env = KuramotoSivashinskyEnv(**_with_device(config))
    return TimeLimit(env, env.unwrapped.max_episode_steps, new_step_api=new_step_api)


def make_sb3(config={}):
    from pdegym.common.wrappers import UnFlattenActionWrapper, UnFlattenObsWrapper
    env = KuramotoSivashinskyEnv(**_with_device(config))
    env = UnFlattenActionWrapper(UnFlattenObsWrapper(env))
    rescale = getattr(gym.wrappers, "RescaleAction", None)
    if rescale is not None:
        env = rescale(env, -1.0, 1.0)
    return TimeLimit(env, env.unwrapped.max_episode_steps)


def make_vec(num_envs, config={}, device=0, devices=None, **kwargs):
    """Batched replacement for ``gym.vector.make(ENV_ID, num_envs=...)``: one GPU batch, or -- ``devices`` = a list of
    ordinals / ``"all"`` -- contiguous env blocks on several GPUs behind one vector env (one controller process)."""
    if devices is not None:
        return KSShardedVecEnv(num_envs, config=config, devices=devices, **kwargs)
    return KSBatchedVecEnv(num_envs, config=config, device=device, **kwargs)


gym.envs.register(id=ENV_ID, entry_point="pdegym.kuramoto:make", order_enforce=False, new_step_api=True)
gym.envs.register(id=ENV_ID_SB3, entry_point="pdegym.kuramoto:make_sb3", order_enforce=False)
