"""Curriculum schedulers for the extrapolation length K (mirror of the reference's
``pdecontrol/surrogates/common/schedulers.py``: ``Scheduler`` :6-16, ``LinearScheduler`` :19-35,
``StepScheduler`` :38-47, ``FuncScheduler`` :50-57, ``ConstantLengthScheduler`` :60-66)."""
import importlib

import numpy as np


class Scheduler:
    """``steptype`` names which of (iteration, epoch, step) drives the schedule."""

    def __init__(self, steptype: str):
        self.steptype = steptype

    def get_step(self, iteration, epoch, step):
        return {"iteration": iteration, "epoch": epoch, "step": step}.get(self.steptype)

    @staticmethod
    def factory(config):
        module = importlib.import_module("pdecontrol.surrogates.common.schedulers")
        return getattr(module, config["scheduler"])(**config)


class LinearScheduler(Scheduler):
    """vmin before ``start``, linear ramp to vmax at ``stop``, clipped."""

    def __init__(self, steptype: str, start: int, stop: int, vmin: float, vmax: float, **kwargs):
        super().__init__(steptype=steptype)
        assert start < stop
        self.start, self.stop, self.vmin, self.vmax = start, stop, vmin, vmax

    def __call__(self, iteration=None, epoch=None, step=None):
        at = self.get_step(iteration, epoch, step)
        fraction = max((at - self.start) / (self.stop - self.start), 0.0)
        return np.clip(self.vmin + fraction * (self.vmax - self.vmin), self.vmin, self.vmax)


class StepScheduler(Scheduler):
    def __init__(self, steptype: str, steps, values, **kwargs):
        super().__init__(steptype=steptype)
        self.steps, self.values = steps, values

    def __call__(self, iteration=None, epoch=None, step=None):
        at = self.get_step(iteration, epoch, step)
        return self.values[np.searchsorted(self.steps, at, side="left")]


class FuncScheduler(Scheduler):
    def __init__(self, steptype: str, func, **kwargs):
        super().__init__(steptype=steptype)
        self.func = func

    def __call__(self, iteration=None, epoch=None, step=None):
        return self.func(self.get_step(iteration, epoch, step))


class ConstantLengthScheduler(Scheduler):
    def __init__(self, length: int, **kwargs):
        super().__init__(steptype="iteration")
        self.length = length

    def __call__(self, iteration=None, epoch=None, step=None):
        return self.length
