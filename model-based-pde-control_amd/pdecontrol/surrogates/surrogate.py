"""Rollout drivers of the PDE surrogate.

API mirror of the reference's ``pdecontrol/surrogates/surrogate.py``: ``PDESurrogate`` :15-19,
``PDEEnsemble`` :22-55, ``AutoRegPDESurrogate`` :58-133 (the hot path: encode -> transition ->
decode a scaled delta -> integrate -> re-encode, teacher forced on the given states then free
running), ``LatentAutoRegPDESurrogate`` :136-206.

Device note: the time-grid / target index tensors (the integer path: ``searchsorted`` and
``round``) are built on the CPU exactly as the reference does and only used as Python index
lists, so they neither sync the GPU nor leak CPU tensors into device arithmetic.
"""
from abc import ABC, abstractmethod
from typing import List

import numpy as np
import torch
from torch import nn

from pdecontrol.mbrl.types import ModelRollout
from pdecontrol.surrogates import ops, utils
from pdecontrol.surrogates.transition import TransitionModel
from pdegym.common.transforms import BatchTransform, Identity


def action_and_target_indices(times, targets, delta):
    """The integer path of a rollout (surrogate.py:85-89,126 in the reference).

    Returns (aidx, tidx): ``aidx[k]`` is the index of the action applied at internal time point k
    (time points ``arange(t0, t_last + delta, delta)``; last action whose time <= the point), and
    ``tidx[j] = round(targets[j] / delta) - 1`` selects the internal step reported for target j."""
    times = torch.as_tensor(times).detach().cpu().reshape(-1)
    targets = torch.as_tensor(targets).detach().cpu().reshape(-1)
    timepoints = torch.arange(times[0], times[-1] + delta, delta)
    aidx = torch.searchsorted(times, timepoints, right=True) - 1
    tidx = torch.round(targets / delta).to(torch.long) - 1
    return aidx, tidx


_INDEX_CACHE = {}


def take_steps(x, idx):
    """``x[:, idx]`` for a Python index list, without a host->device copy on the hot path: a
    contiguous ascending run becomes a slice (the common case: one action / one target per step);
    anything else goes through ``index_select`` with a per-(indices, device) cached index tensor,
    created once (e.g. during graph warm-up) so HIP-graph capture never sees an H2D copy."""
    idx = [int(i) for i in idx]
    if idx == list(range(idx[0], idx[0] + len(idx))) and idx[0] >= 0:
        return x[:, idx[0]:idx[0] + len(idx)]
    key = (tuple(idx), x.device)
    if key not in _INDEX_CACHE:
        _INDEX_CACHE[key] = torch.tensor(idx, dtype=torch.long, device=x.device)
    return torch.index_select(x, 1, _INDEX_CACHE[key])


class PDESurrogate(ABC, nn.Module):
    @abstractmethod
    def rollout(self, states, actions, times, targets) -> ModelRollout:
        pass


class PDEEnsemble(PDESurrogate):
    """Runs every member, then picks for each batch row the prediction of a random elite."""

    def __init__(self, modules: List, num_elites: int = None):
        super().__init__()
        self.modules = modules
        self.num_elites = len(modules) if num_elites is None else num_elites
        self.elite_idx: List[int] = list(range(len(modules)))

    def rollout(self, states, actions, times, targets, hidden=None) -> ModelRollout:
        hidden = [None] * len(self.modules) if hidden is None else hidden
        rollouts = [m.surrogate.rollout(states, actions, times, targets, hidden=h)
                    for m, h in zip(self.modules, hidden)]
        chosen = np.random.choice(self.elite_idx, size=states.size(0))
        stacked = torch.stack([r.outputs for r in rollouts], dim=0)               # [M, B, ...]
        rows = torch.arange(states.size(0), device=stacked.device)
        outputs = stacked[torch.as_tensor(chosen, device=stacked.device), rows]   # gather, no Python loop
        return ModelRollout(outputs=outputs, hidden=[r.hidden for r in rollouts])

    def update_elites(self, scores: List[float]) -> None:
        self.elite_idx = list(np.argsort(scores)[:self.num_elites])


class _EncDecSurrogate(PDESurrogate):
    def __init__(self, state_encoder: nn.Module, state_decoder: nn.Module, action_encoder: nn.Module,
                 transition_model: TransitionModel, delta: float, dscaling: BatchTransform = None, **kwargs):
        super().__init__()
        self.delta = delta
        self.dscaling = BatchTransform(Identity()) if dscaling is None else dscaling
        self.state_encoder = utils.BatchingWrapper(state_encoder)
        self.state_decoder = utils.BatchingWrapper(state_decoder)
        self.action_encoder = utils.BatchingWrapper(action_encoder)
        self.transition_model = transition_model


class AutoRegPDESurrogate(_EncDecSurrogate):
    """State-space autoregression: next = prev + delta * dscaling(decoder(transition(...))).

    ``reencode_predictions``: the reference re-encodes every prediction (``inlast``) but its
    free-running transition ignores that input (transition.py:285-296), so the encoder pass only
    ever populates ``ModelRollout.inlatents``, which nothing reads.  True (default) keeps that
    field identical to the reference; the training module switches it off (inlatents = None) and
    saves roughly a third of the forward work.  Outputs / deltas / hidden are unaffected."""

    reencode_predictions = True

    def rollout(self, states: torch.Tensor, actions: torch.Tensor, times: torch.Tensor, targets: torch.Tensor,
                hidden=None, **kwargs) -> ModelRollout:
        if ops.use_fused_for(self, states):
            # one launch per module instead of ~60 per time step (no inlatents in this mode)
            from pdecontrol.surrogates import hipops
            return hipops.fused_rollout(self, states, actions, times, targets, hidden)
        n_given = states.size(1)
        lstates = self.state_encoder(states)
        aidx, tidx = action_and_target_indices(times, targets, self.delta)
        lactions = take_steps(self.action_encoder(actions), aidx.tolist())
        n_steps = lactions.size(1)

        inlatents, outlatents, outdeltas, outputs = [], [], [], []
        output, inlast = states[:, :1], lstates[:, :1]
        for k in range(n_steps):
            laction = lactions[:, k:k + 1]
            if k < n_given:   # teacher forcing on the given (warm-up) states
                inlatent, base = lstates[:, k:k + 1], states[:, k:k + 1]
                outlatent, hidden = self.transition_model.teacherforcing(states=inlatent, actions=laction,
                                                                         hidden=hidden, **kwargs)
            else:             # free running on the model's own (detached) re-encoded prediction
                inlatent, base = inlast, output
                outlatent, hidden = self.transition_model.transition(states=inlatent, actions=laction,
                                                                     hidden=hidden, **kwargs)
            outdelta = self.state_decoder(outlatent)
            output = base + self.delta * self.dscaling(outdelta)
            # the re-encoded prediction only feeds a *following free-running* step (the reference
            # re-encodes after every step and discards most of them; values are identical)
            if self.reencode_predictions and n_given <= k + 1 < n_steps:
                inlast = self.state_encoder(output).detach()
            inlatents.append(inlatent)
            outlatents.append(outlatent)
            outdeltas.append(outdelta)
            outputs.append(output)

        pick = tidx.tolist()
        gather = lambda seq: take_steps(torch.cat(seq, dim=1), pick)
        return ModelRollout(inlatents=gather(inlatents) if self.reencode_predictions else None,
                            outlatents=gather(outlatents), deltas=gather(outdeltas), outputs=gather(outputs),
                            hidden=hidden)


class LatentAutoRegPDESurrogate(_EncDecSurrogate):
    """Latent-space autoregression: the latent integrates, the decoder reads it out."""

    def rollout(self, states: torch.Tensor, actions: torch.Tensor, times: torch.Tensor, targets: torch.Tensor,
                hidden=None, **kwargs) -> ModelRollout:
        ops.require_plain_path(states, "LatentAutoRegPDESurrogate (ablation)")
        n_given = states.size(1)
        lstates = self.state_encoder(states)
        aidx, tidx = action_and_target_indices(times, targets, self.delta)
        lactions = take_steps(self.action_encoder(actions), aidx.tolist())

        inlatents, outlatents, outputs = [], [], []
        inlatent = lstates[:, :1]
        for k in range(lactions.size(1)):
            laction = lactions[:, k:k + 1]
            if k < n_given:
                inlatents.append(lstates[:, k:k + 1])
                outlatent, hidden = self.transition_model.teacherforcing(states=lstates[:, k:k + 1], actions=laction,
                                                                         hidden=hidden, **kwargs)
            else:
                inlatents.append(inlatent)
                outlatent, hidden = self.transition_model.transition(states=inlatent, actions=laction,
                                                                     hidden=hidden, **kwargs)
            inlatent = inlatent + self.delta * outlatent
            outlatents.append(outlatent)
            outputs.append(self.state_decoder(inlatent))

        outputs = torch.cat(outputs, dim=1)
        deltas = self.dscaling.Inverse(torch.diff(torch.cat((states[:, :1], outputs), dim=1), dim=1) / self.delta)
        pick = tidx.tolist()
        return ModelRollout(inlatents=take_steps(torch.cat(inlatents, dim=1), pick),
                            outlatents=take_steps(torch.cat(outlatents, dim=1), pick), deltas=deltas,
                            outputs=take_steps(outputs, pick), hidden=hidden)
