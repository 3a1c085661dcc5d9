"""Ensemble-parallel TBPTT: all members of the dynamics ensemble take their optimizer step at once.

The reference updates its ensemble one member after the other (pdecontrol/mbrl/mbrl.py:408: a list
comprehension over ``update_surrogate(module, trainer)``; 3 members by default, script.py:60), each on
its own bootstrapped batches.  On an MI355X one member's fused step is 64 workgroups (one per sequence)
on a 256-CU device and bound by the dependent-phase latency inside each workgroup, so the members do
not compete for anything: every member's captured step graph is replayed on its own HIP stream and the
hardware runs them side by side.  Results are bit-identical to stepping the members in turn (same
graphs, same kernels, deterministic gradient reduction).
"""
from typing import Sequence

import torch

from pdecontrol.surrogates.graph_step import GraphedTBPTTStep


class EnsembleTBPTTStep:
    def __init__(self, modules: Sequence, batch_shape, action_shape=None, lr=None, warmup=3):
        """modules: PDETrainingModules on ONE CUDA device (distinct parameters); batch_shape: [B, T, 1, N]."""
        assert len(modules) > 0
        self.members = [GraphedTBPTTStep(m, batch_shape, action_shape, lr=lr, warmup=warmup) for m in modules]
        self.device = self.members[0].device
        assert all(g.device == self.device for g in self.members), "ensemble members must share one GPU"
        self.streams = [torch.cuda.Stream(device=self.device) for _ in self.members]

    def __len__(self):
        return len(self.members)

    def load(self, batches):
        """batches: one (states, actions) pair per member (bootstrapped batches differ per member)."""
        assert len(batches) == len(self.members)
        for g, (s, a) in zip(self.members, batches):
            g.states.copy_(s, non_blocking=True)
            g.actions.copy_(a, non_blocking=True)

    def step(self, batches=None):
        """One optimizer step of every member, concurrently.  Returns the members' static result dicts;
        the caller's current stream waits for all of them."""
        cur = torch.cuda.current_stream(self.device)
        if batches is not None:
            self.load(batches)
        for g, st in zip(self.members, self.streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                g.g_main.replay()
        for st in self.streams:
            cur.wait_stream(st)
        return [g.result for g in self.members]
