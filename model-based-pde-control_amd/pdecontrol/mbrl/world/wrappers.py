"""Hook point of the world model inside the real-env wrapper stack (mirror of the reference's
``pdecontrol/mbrl/world/wrappers.py:13-47``): a pass-through that carries the surrogate, its time step
and optional callbacks, and can be disabled."""
from typing import Any, List, Sequence

from pdegym._gym import gym


class BaseWorldVecEnvWrapper(gym.vector.VectorEnvWrapper):
    def __init__(self, env, surrogate, tstep: float, callbacks: List = None):
        super().__init__(env)
        self.surrogate, self.tstep = surrogate, tstep
        self.callbacks = [] if callbacks is None else callbacks
        self._enabled = True

    def step_wait(self, **kwargs: Any):
        return self.env.step_wait(**kwargs)

    def step_async(self, actions: Sequence[Any]) -> None:
        return self.env.step_async(actions)

    def reset(self, **kwargs) -> Any:
        return self.env.reset(**kwargs)

    def disable(self) -> None:
        self._enabled = False

    def enable(self) -> None:
        self._enabled = True
