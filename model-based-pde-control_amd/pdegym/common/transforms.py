"""Invertible value transforms shared by the env, the wrappers and the surrogate.

API mirror of the reference's ``pdegym/common/transforms.py`` (same class names, constructor
arguments, attributes and ``.Inverse`` protocol), re-implemented around two tensor hooks
(``_fwd`` / ``_inv``) with one numpy<->torch conversion point, and device-aware: statistics and
matrices follow the device/dtype of the tensor they are applied to, so the same objects serve CPU
numpy data at the gym boundary and HBM-resident batches inside the TBPTT step.

Reference behaviour pinned by tests/golden (tests/test_transforms.py):
  Normalize        transforms.py:62-138   (running mean / population-corrected variance merge)
  ScaleTransform   transforms.py:141-210
  FuncTransform    transforms.py:213-228
  SensorTransform  transforms.py:231-247
  GaussianForcing  transforms.py:250-279  (fp32, non-periodic Gaussians, 1/sqrt(2 pi sigma))
  BatchTransform / Operation / SampleTransform  transforms.py:282-374
"""
from typing import Iterable, List, Sequence

import numpy as np
import torch


def _to_tensor(values):
    """numpy -> (tensor view, True); tensor -> (tensor, False)."""
    if isinstance(values, np.ndarray):
        return torch.from_numpy(values), True
    return values, False


def _from_tensor(t, was_numpy):
    return t.detach().numpy() if was_numpy else t


_DEVICE_COPIES = {}


def _on(stat, like):
    """Statistic tensor on the device of ``like``.  Device copies are cached per (source tensor,
    device) -- the source is kept alive in the cache entry, so an entry can only be hit by the very
    tensor it was made from; re-assigned statistics (``update`` always re-assigns) miss and are
    copied afresh.  This keeps host->device copies out of steady-state steps (and therefore out of
    HIP-graph capture, where they are illegal)."""
    if not (isinstance(stat, torch.Tensor) and isinstance(like, torch.Tensor)) or stat.device == like.device:
        return stat
    key = (id(stat), like.device)
    hit = _DEVICE_COPIES.get(key)
    if hit is None or hit[0] is not stat:
        if len(_DEVICE_COPIES) > 256:
            _DEVICE_COPIES.clear()
        hit = (stat, stat.to(like.device))
        _DEVICE_COPIES[key] = hit
    return hit[1]


class _InverseView:
    """``T.Inverse``: applies ``T._inv``; ``update`` maps the values back through the inverse and
    feeds the forward transform's statistics; ``.Inverse`` returns the forward transform."""

    def __init__(self, transf: "Transform"):
        self.transf = transf

    def __call__(self, *values):
        return self.transf._apply(self.transf._inv, *values)

    def update(self, values):
        self.transf.update(self(values))

    @property
    def Inverse(self):
        return self.transf


class Transform:
    #: True when the transform acts independently of leading batch dimensions, so that
    #: BatchTransform may apply it to the whole batch at once instead of item by item.
    batchable = False
    _Inverse = _InverseView

    # -- hooks ------------------------------------------------------------------------------
    def _fwd(self, t):
        raise NotImplementedError

    def _inv(self, t):
        raise NotImplementedError

    # -- protocol ---------------------------------------------------------------------------
    def _apply(self, fn, *values):
        first_numpy = isinstance(values[0], np.ndarray)
        tensors = [_to_tensor(v)[0] for v in values]
        return _from_tensor(fn(*tensors), first_numpy)

    def __call__(self, values):
        return self._apply(self._fwd, values)

    def update(self, values):
        pass

    @property
    def Inverse(self):
        return self._Inverse(self)

    # kept for callers that used the reference's helpers directly
    @staticmethod
    def convert(values):
        t, was_numpy = _to_tensor(values)
        return t, (np.ndarray if was_numpy else torch.Tensor)

    @staticmethod
    def unconvert(values, vtype):
        return _from_tensor(values, vtype == np.ndarray)


class Identity(Transform):
    batchable = True

    def _fwd(self, t):
        return t

    _inv = _fwd


def _reduce_dims(aggregate, batched):
    if aggregate and batched:
        return (0, 1, 2)
    if aggregate or batched:
        return (0, 1)
    return (0,)


class Normalize(Transform):
    """Running standardisation; statistics are kept as fp32 tensors with keepdim shapes."""
    batchable = True

    def __init__(self, aggregate=False, batched=False, frozen=False, epsilon=1e-4):
        self.aggregate, self.batched, self.frozen, self.epsilon = aggregate, batched, frozen, epsilon
        self.dim = _reduce_dims(aggregate, batched)
        self.reset()

    def reset(self):
        self.mean, self.var, self.count = None, None, 0

    def _fwd(self, t):
        return (t - _on(self.mean, t)) / torch.sqrt(_on(self.var, t) + self.epsilon)

    def _inv(self, t):
        return t * torch.sqrt(_on(self.var, t) + self.epsilon) + _on(self.mean, t)

    def update(self, values):
        if self.frozen:
            return
        t, _ = _to_tensor(values)
        n_new = t.shape[0]
        b_mean = torch.mean(t, dim=self.dim, keepdim=True, dtype=torch.float32)
        b_var = torch.var(t, dim=self.dim, keepdim=True)
        if self.mean is None:
            self.mean = torch.zeros_like(b_mean)
        if self.var is None:
            self.var = torch.zeros_like(b_mean)
        mean, var = _on(self.mean, t), _on(self.var, t)
        total = self.count + n_new
        delta = b_mean - mean
        # parallel-variance merge (Chan et al.), with the batch size as the weight of both terms
        m2 = var * self.count + b_var * n_new + delta * delta * self.count * n_new / total
        self.mean = mean + delta * n_new / total
        self.var = m2 / total
        self.count = total


class ScaleTransform(Transform):
    """Affine map of [vmin, vmax] (tracked running extrema) onto ``scale``."""
    batchable = True

    def __init__(self, scale=(-1.0, 1.0), bounds=(-np.inf, np.inf), aggregate=False, batched=False, frozen=False):
        self.aggregate, self.batched, self.frozen = aggregate, batched, frozen
        self.dim = _reduce_dims(aggregate, batched)
        f32 = lambda v: torch.from_numpy(np.asarray(v, dtype=np.float32))
        self.lower, self.upper = f32(scale[0]), f32(scale[1])
        self.vmin, self.vmax = f32(bounds[0]), f32(bounds[1])
        if aggregate and self.vmin.ndim > 1 and self.vmax.ndim > 1:
            self.vmin = torch.amin(self.vmin, dim=self.dim, keepdim=True)
            self.vmax = torch.amax(self.vmax, dim=self.dim, keepdim=True)

    # Same expression and operation order as the reference ((v - a) / (b - a) * (d - c) + c, every step rounded in the
    # operands' precision), written with in-place steps (one temporary per batch instead of four).  numpy batches -- the
    # gym-boundary path of the vector wrappers -- are handled in numpy: IEEE arithmetic is the same, and torch's CPU
    # ops fan tiny [E, 1, N] batches out over every OpenMP thread of the box (0.4 ms per op on a 128-thread host).
    @staticmethod
    def _affine(t, a, b, c, d):
        out = t - a
        # in place only for >= 1-D results whose dtype the division keeps (0-d inputs -- torch scalars, np.float32 --
        # cannot be sliced for the probe and gain nothing from it)
        if getattr(out, "ndim", 0) > 0 and out.shape == t.shape and out.dtype == (out[:0] / (b - a)).dtype:
            out /= (b - a)
            out *= (d - c)
            out += c
            return out
        return out / (b - a) * (d - c) + c

    def _fwd(self, t):
        vmin, vmax = _on(self.vmin, t), _on(self.vmax, t)
        lower, upper = _on(self.lower, t), _on(self.upper, t)
        return self._affine(t, vmin, vmax, lower, upper)

    def _inv(self, t):
        vmin, vmax = _on(self.vmin, t), _on(self.vmax, t)
        lower, upper = _on(self.lower, t), _on(self.upper, t)
        return self._affine(t, lower, upper, vmin, vmax)

    def _apply(self, fn, *values):
        v = values[0]
        if len(values) == 1 and isinstance(v, np.ndarray) and not self.vmin.is_cuda:
            vmin, vmax, lower, upper = (x.numpy() for x in (self.vmin, self.vmax, self.lower, self.upper))
            with np.errstate(invalid="ignore", divide="ignore"):   # unset (+-inf) bounds give nan, as torch does, silently
                if fn == self._fwd:
                    return self._affine(v, vmin, vmax, lower, upper)
                if fn == self._inv:
                    return self._affine(v, lower, upper, vmin, vmax)
        return super()._apply(fn, *values)

    def update(self, values):
        if self.frozen:
            return
        if isinstance(values, np.ndarray) and len(self.dim) == values.ndim and not self.vmin.is_cuda:
            shape = (1,) * values.ndim       # extrema over every axis, in numpy (see above)
            lo = torch.from_numpy(np.asarray(values.min(), dtype=values.dtype).reshape(shape))
            hi = torch.from_numpy(np.asarray(values.max(), dtype=values.dtype).reshape(shape))
            t = lo
        else:
            t, _ = _to_tensor(values)
            lo = torch.amin(t, dim=self.dim, keepdim=True)
            hi = torch.amax(t, dim=self.dim, keepdim=True)
        # an unset bound (+-inf scalar) is replaced by the first batch's extremum
        self.vmin = lo if bool(torch.all(torch.isneginf(self.vmin))) else torch.minimum(lo, _on(self.vmin, t))
        self.vmax = hi if bool(torch.all(torch.isposinf(self.vmax))) else torch.maximum(hi, _on(self.vmax, t))


class FuncTransform(Transform):
    """Wraps arbitrary callables (the env's reward function); accepts several arguments."""

    def __init__(self, transf, inverse=None):
        self.transf, self.inverse = transf, inverse

    def _fwd(self, *tensors):
        return self.transf(*tensors)

    def _inv(self, *tensors):
        return self.inverse(*tensors)

    def __call__(self, *args):
        return self._apply(self._fwd, *args)


class SensorTransform(Transform):
    """Every ``stride``-th grid point, starting at ``stride // 2``."""
    batchable = True

    def __init__(self, stride):
        self.stride = stride

    def _fwd(self, t):
        return t[..., int(self.stride / 2)::self.stride]

    def _inv(self, t):
        if self.stride > 1:
            raise NotImplementedError()
        return t


class _ForcingInverse(_InverseView):
    """pattern -> action: sample the pattern at the actuator grid points, solve the 4x4 system."""

    def __init__(self, transf):
        super().__init__(transf)
        self.xpos = (transf.N * transf.Xi.reshape(-1, 1)).to(dtype=torch.long).reshape(-1)
        self.inv_forcing = torch.inverse(transf.forcing[:, self.xpos])

    def __call__(self, values):
        def inv(t):
            return t[..., _on(self.xpos, t)] @ _on(self.inv_forcing, t)
        return self.transf._apply(inv, values)


class GaussianForcing(Transform):
    """action [..., n_act] -> forcing field [..., N]: ``values @ forcing`` in fp32, where
    forcing[j, i] = exp(-(x_i - L*Xi_j)^2 / (2 sigma^2)) / sqrt(2 pi sigma)  (not periodic)."""
    batchable = True
    _Inverse = _ForcingInverse

    def __init__(self, x: Sequence, Xi: Sequence, sigma: float, L: float, N: int):
        self.sigma, self.L, self.N = sigma, L, N
        self.x = torch.as_tensor(x).to(dtype=torch.float32)
        self.Xi = torch.as_tensor(Xi).to(dtype=torch.float32)
        self.xi = (self.L * self.Xi).reshape(-1, 1)
        gauss = torch.exp(-((self.x - self.xi) ** 2.0) / (2.0 * sigma ** 2))
        self.forcing = gauss / np.sqrt(2.0 * np.pi * self.sigma)

    def _fwd(self, t):
        return t @ _on(self.forcing, t)


class _BatchInverse(_InverseView):
    def __init__(self, transf):
        super().__init__(transf)
        self.transform = transf.transform.Inverse

    def __call__(self, values):
        return self.transf._map(self.transform, values)


class BatchTransform(Transform):
    """Applies ``transform`` to every item along dim 0.  Batchable inner transforms are applied to
    the whole batch in one call (same arithmetic, no Python loop); others item by item."""
    _Inverse = _BatchInverse

    def __init__(self, transform: Transform):
        self.transform = transform

    def _map(self, fn, values):
        t, was_numpy = _to_tensor(values)
        inner = fn.transf if isinstance(fn, _InverseView) else fn
        if getattr(inner, "batchable", False):
            out = fn(t)
        else:
            out = torch.stack([fn(item) for item in t], dim=0)
        return _from_tensor(out, was_numpy)

    def __call__(self, values):
        return self._map(self.transform, values)

    def update(self, values):
        inner = self.transform
        if type(inner).update is Transform.update:
            return                      # stateless inner transform: nothing to update (and no per-item loop)
        t, _ = _to_tensor(values)
        if isinstance(inner, ScaleTransform):
            # running min / max are exact and order-independent: reduce every item at once, then fold
            # the per-item extrema -- same result as updating item by item, without the Python loop
            if inner.frozen or t.shape[0] == 0:
                return
            dims = tuple(d + 1 for d in inner.dim)
            lo = torch.amin(torch.amin(t, dim=dims, keepdim=True), dim=0)
            hi = torch.amax(torch.amax(t, dim=dims, keepdim=True), dim=0)
            inner.vmin = lo if bool(torch.all(torch.isneginf(inner.vmin))) else torch.minimum(lo, _on(inner.vmin, t))
            inner.vmax = hi if bool(torch.all(torch.isposinf(inner.vmax))) else torch.maximum(hi, _on(inner.vmax, t))
            return
        for item in t:                  # order-dependent running statistics (Normalize): item by item
            inner.update(item)


class _OperationInverse(_InverseView):
    def __init__(self, transf):
        super().__init__(transf)
        self.transfs = [t.Inverse for t in reversed(transf.transforms)]

    def __call__(self, values):
        t, was_numpy = _to_tensor(values)
        for fn in self.transfs:
            t = fn(t)
        return _from_tensor(t, was_numpy)


class Operation(Transform):
    """Composition, applied left to right."""
    _Inverse = _OperationInverse

    def __init__(self, transforms: List[Transform]):
        self.transforms = transforms

    def __call__(self, values):
        t, was_numpy = _to_tensor(values)
        for fn in self.transforms:
            t = fn(t)
        return _from_tensor(t, was_numpy)

    def update(self, values):
        t, _ = _to_tensor(values)
        for fn in self.transforms:
            fn.update(t)
            t = fn(t)
        return t


class SampleTransform(Transform):
    """Observation / action transforms applied to a replay ``Sample``."""

    def __init__(self, otransf=None, atransf=None):
        def as_operation(tr):
            if tr is None:
                tr = BatchTransform(Identity())
            return Operation(list(tr) if isinstance(tr, Iterable) else [tr])
        self.otransf, self.atransf = as_operation(otransf), as_operation(atransf)

    def __call__(self, sample):
        from pdecontrol.mbrl.types import Sample
        obs, actions, nxtobs, rewards, terminated, truncated, steps = sample
        return Sample(self.otransf(obs), self.atransf(actions), self.otransf(nxtobs), rewards, terminated,
                      truncated, steps)

    @property
    def Inverse(self):
        return SampleTransform(otransf=self.otransf.Inverse, atransf=self.atransf.Inverse)
