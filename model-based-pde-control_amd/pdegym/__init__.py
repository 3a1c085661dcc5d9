"""pdegym -- PDE control environments (MI355X-native build).

Importing the package registers the environments with gym, like the reference's
``pdegym/__init__.py`` -- including ``pdegym.burgers``, which the reference imports but does not ship
(its ``import pdegym`` fails as shipped): here that module exists (pdegym/burgers).

Batched GPU vector envs: set ``PDEGYM_BATCHED=1`` (or call ``install_batched_vector_make()``) and
``gym.vector.make("KuramotoSivashinskyEnv-v0", num_envs=E)`` -- the call the reference's
controller makes (pdecontrol/mbrl/mbrl.py:81-86) -- returns one ``KSBatchedVecEnv`` instead of E
subprocess envs, so ``pdecontrol/mbrl/script.py`` needs no edit.  With ``PDEGYM_DEVICES=all`` (or ``0,1,2,3``) the same
call returns a ``KSShardedVecEnv``: the one controller process drives every listed GPU.
"""
import os

from pdegym._gym import gym, IS_SHIM
import pdegym.burgers  # noqa: F401  (registers BurgersEnv-v0; loads no GPU library until an env is built)
import pdegym.kuramoto  # noqa: F401  (registers the env ids)


def install_batched_vector_make(device=None, devices=None):
    """Route ``gym.vector.make`` for the KS env ids to the batched HIP vector env (``devices``: to the sharded one)."""
    from pdegym.kuramoto import ENV_ID, make_vec
    previous = getattr(gym.vector, "make", None)
    if getattr(previous, "_pdegym_batched", False):
        return

    def vector_make(id, num_envs=1, asynchronous=True, wrappers=None, **kwargs):
        if id == ENV_ID:
            kwargs.pop("new_step_api", None)
            devs = devices if devices is not None else (os.environ.get("PDEGYM_DEVICES") or None)
            if devs is not None:
                return make_vec(num_envs, config=kwargs.pop("config", {}), devices=devs, **kwargs)
            dev = device if device is not None else int(os.environ.get("LOCAL_RANK", "0"))
            return make_vec(num_envs, config=kwargs.pop("config", {}), device=dev, **kwargs)
        if previous is None:
            raise KeyError(f"no vector factory for env {id!r}")
        return previous(id, num_envs=num_envs, asynchronous=asynchronous, wrappers=wrappers, **kwargs)

    vector_make._pdegym_batched = True
    gym.vector.make = vector_make


if os.environ.get("PDEGYM_BATCHED", "0") not in ("", "0"):
    install_batched_vector_make()
