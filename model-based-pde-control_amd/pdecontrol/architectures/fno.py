"""FNO-style autoregressive surrogate for the Burgers path (BASELINE configs[4]).

Nothing to mirror -- the reference has neither a Burgers env nor an FNO.  The pieces follow the reference's
own conventions so that they plug into its training module and world env unchanged:

* ``FNO1d``: lift (pointwise conv: [state, action field] -> width) -> ``layers`` x [SpectralConv1d + pointwise conv, GELU]
  -> project (pointwise conv width -> width -> 1); the published architecture (Li et al. 2021) in 1-D.
* ``FNOAutoRegSurrogate``: the same contract as ``AutoRegPDESurrogate.rollout`` (pdecontrol/surrogates/surrogate.py:79-133):
  ``next = prev + delta * dscaling(model(prev, action))``, teacher forced on the given states, free running afterwards,
  same integer path for action / target indices; stateless (``hidden`` is an empty tuple).  ``training_mode = "delta"``.
* ``BurgersFNO``: the factory (``--factory BurgersFNO``), ``model(N=..., width=..., modes=..., layers=...)``.

On CUDA tensors a rollout of the default geometry (width 32, 16 modes, 4 layers, N <= 512) runs on the whole-network HIP
kernels (pdecontrol/surrogates/fno_hip.py, csrc/fno.hip): one launch per model evaluation and direction instead of ~170
small kernels; any other geometry keeps the per-operator path below (fused spectral convolution + torch / rocBLAS).
"""
import torch
from torch import nn

from pdecontrol.mbrl.types import ModelRollout
from pdecontrol.surrogates.factory import PDESurrogateFactory
from pdecontrol.surrogates.spectral import SpectralConv1d
from pdecontrol.surrogates.surrogate import PDESurrogate, action_and_target_indices, take_steps
from pdegym.common.transforms import BatchTransform, Identity


class _PointwiseFn(torch.autograd.Function):
    """y = W x + b over [B, C, N] with a backward that keeps rocBLAS on well-shaped GEMMs: autograd's default weight
    gradient contracts over B * N at once -- ONE 32 x 32 output tile with K = 32 768, a single workgroup grinding for
    0.2 ms (41 % of the eager FNO step in profiles/r02_fno_kernel_stats_a.txt).  Here it is a batched GEMM per sample
    (B workgroups, K = N) followed by a sum over the batch."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.matmul(weight, x) + bias[:, None]

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = torch.matmul(weight.t(), dy) if ctx.needs_input_grad[0] else None
        dw = torch.bmm(dy, x.transpose(1, 2)).sum(0) if ctx.needs_input_grad[1] else None
        db = dy.sum(dim=(0, 2)) if ctx.needs_input_grad[2] else None
        return dx, dw, db


class Pointwise(nn.Module):
    """Channel mixing at every grid point: y[b, o, n] = sum_i W[o, i] x[b, i, n] + bias[o] -- a plain (batched) GEMM,
    spelled as one so that it runs on rocBLAS rather than through a convolution library.  Same initialisation as
    ``nn.Conv1d(cin, cout, 1)``."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        conv = nn.Conv1d(in_channels, out_channels, 1)
        self.weight = nn.Parameter(conv.weight.detach().squeeze(-1).clone())
        self.bias = nn.Parameter(conv.bias.detach().clone())

    def forward(self, x):
        if x.dim() == 3 and x.is_cuda:
            if min(self.weight.shape) <= 2:
                # lift (2 -> width) and the last projection (width -> 1): a GEMM with an extent of 1 or 2 sends rocBLAS to
                # 0.1 ms kernels; as broadcast multiply + reduce it is two small elementwise launches
                return (self.weight[None, :, :, None] * x[:, None, :, :]).sum(dim=2) + self.bias[:, None]
            return _PointwiseFn.apply(x, self.weight, self.bias)
        return torch.matmul(self.weight, x) + self.bias[:, None]


class FNO1d(nn.Module):
    def __init__(self, in_channels: int = 2, width: int = 32, modes: int = 16, layers: int = 4):
        super().__init__()
        self.lift = Pointwise(in_channels, width)
        self.spectral = nn.ModuleList([SpectralConv1d(width, width, modes) for _ in range(layers)])
        self.pointwise = nn.ModuleList([Pointwise(width, width) for _ in range(layers)])
        self.project = nn.Sequential(Pointwise(width, width), nn.GELU(), Pointwise(width, 1))
        self.activation = nn.GELU()

    def forward(self, x):                                   # [M, in_channels, N] -> [M, 1, N]
        x = self.lift(x)
        last = len(self.spectral) - 1
        for k, (spec, pw) in enumerate(zip(self.spectral, self.pointwise)):
            x = spec(x) + pw(x)
            if k < last:
                x = self.activation(x)
        return self.project(x)


class FNOAutoRegSurrogate(PDESurrogate):
    training_mode = "delta"

    def __init__(self, model: nn.Module, delta: float, dscaling: BatchTransform = None, **kwargs):
        super().__init__()
        self.model, self.delta = model, delta
        self.dscaling = BatchTransform(Identity()) if dscaling is None else dscaling

    def rollout(self, states, actions, times, targets, hidden=None, **kwargs) -> ModelRollout:
        n_given = states.size(1)
        aidx, tidx = action_and_target_indices(times, targets, self.delta)
        acts = take_steps(actions, aidx.tolist())
        from pdecontrol.surrogates import ops
        if ops.use_fused(states):
            from pdecontrol.surrogates import fno_hip
            fused = fno_hip.rollout(self.model, states, acts, n_given, self.delta, self.dscaling)
            if fused is not None:
                deltas, outputs = fused
                pick = tidx.tolist()
                return ModelRollout(outlatents=None, deltas=take_steps(deltas, pick), outputs=take_steps(outputs, pick), hidden=())
        outdeltas, outputs = [], []
        output = states[:, 0]
        for k in range(acts.size(1)):
            base = states[:, k] if k < n_given else output          # [B, 1, N]
            outdelta = self.model(torch.cat((base, acts[:, k]), dim=1))
            output = base + self.delta * self.dscaling(outdelta)
            outdeltas.append(outdelta)
            outputs.append(output)
        pick = tidx.tolist()
        gather = lambda seq: take_steps(torch.stack(seq, dim=1), pick)
        return ModelRollout(outlatents=None, deltas=gather(outdeltas), outputs=gather(outputs), hidden=())


class BurgersFNO(PDESurrogateFactory):
    def surrogate(self, model=None, **kwargs):
        return FNOAutoRegSurrogate(model=model, **kwargs)

    def model(self, width: int = 32, modes: int = 16, layers: int = 4, **kwargs):
        return {"model": FNO1d(in_channels=2, width=width, modes=modes, layers=layers)}
