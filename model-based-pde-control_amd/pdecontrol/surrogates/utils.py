"""Small helpers of the surrogate package (API mirror of the reference's
``pdecontrol/surrogates/utils.py``: ``Conv1dDerivative`` :8-32, ``BatchingWrapper`` :35-47,
``ignore_extra_keywords`` :50-61)."""
import functools
import inspect

import torch
from torch import nn


class Conv1dDerivative(nn.Module):
    """Fixed (non-trainable) finite-difference stencil as a Conv1d, divided by ``resolution``."""

    def __init__(self, filter, resolution, kernel_size, padding=0, padding_mode="zeros"):
        super().__init__()
        self.resolution = resolution
        self.filter = nn.Conv1d(1, 1, kernel_size, stride=1, padding=padding, padding_mode=padding_mode, bias=False)
        self.filter.weight = nn.Parameter(torch.as_tensor(filter, dtype=torch.float32), requires_grad=False)

    def forward(self, input):
        return self.filter(input) / self.resolution


class BatchingWrapper(nn.Module):
    """Runs a ``[B, C, H]`` model over ``[B, T, C, H]`` input by folding time into the batch."""

    def __init__(self, model: nn.Module) -> None:
        super().__init__()
        self.model = model

    def forward(self, input):
        b, t = input.shape[:2]
        out = self.model(input.reshape(b * t, *input.shape[2:]))
        return out.reshape(b, t, *out.shape[1:])


def ignore_extra_keywords(func):
    """Drop keyword arguments ``func`` does not declare (no-op if it already takes ``**kwargs``)."""
    params = inspect.signature(func).parameters.values()
    if any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params):
        return func
    accepted = {p.name for p in params if p.kind is not inspect.Parameter.VAR_POSITIONAL}

    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        return func(*args, **{k: v for k, v in kwargs.items() if k in accepted})

    return wrapper
