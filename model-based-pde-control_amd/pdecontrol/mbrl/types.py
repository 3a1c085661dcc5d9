"""Record types exchanged between the env side and the surrogate side.

API mirror of the reference's ``pdecontrol/mbrl/types.py`` (``Sample`` :9-69, ``ModelRollout``
:72-81, ``PDETrainer`` :84-91): same field names and order, iteration yields the fields in order.
"""
from dataclasses import dataclass, fields
from typing import Any, Callable, List

import numpy as np
import torch

_SAMPLE_TORCH = (torch.FloatTensor, torch.FloatTensor, torch.FloatTensor, torch.FloatTensor, torch.BoolTensor,
                 torch.BoolTensor, torch.IntTensor)


class _FieldIter:
    def __iter__(self):
        return iter(tuple(getattr(self, f.name) for f in fields(self)))


@dataclass
class Sample(_FieldIter):
    obs: Any = None
    actions: Any = None
    nxtobs: Any = None
    rewards: Any = None
    terminated: Any = None
    truncated: Any = None
    steps: Any = None

    def totorch(self) -> "Sample":
        for f, ctor in zip(fields(self), _SAMPLE_TORCH):
            setattr(self, f.name, ctor(getattr(self, f.name)))
        return self

    def tonumpy(self) -> "Sample":
        for f in fields(self):
            setattr(self, f.name, getattr(self, f.name).numpy())
        return self

    def apply(self, func: Callable) -> "Sample":
        return Sample(*map(func, self))

    def split(self, axis=0) -> List["Sample"]:
        moved = [np.moveaxis(v, axis, 0) for v in self]
        return [Sample(*row) for row in zip(*moved)]


@dataclass
class ModelRollout(_FieldIter):
    outputs: Any = None
    inlatents: Any = None
    outlatents: Any = None
    deltas: Any = None
    hidden: Any = None


@dataclass
class PDETrainer(_FieldIter):
    trainer: Any = None
    early_stopping: Any = None
