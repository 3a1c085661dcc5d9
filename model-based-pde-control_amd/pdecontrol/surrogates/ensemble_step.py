"""Ensemble-parallel TBPTT: all members of the dynamics ensemble take their optimizer step at once.

The reference updates its ensemble one member after the other (pdecontrol/mbrl/mbrl.py:408: a list
comprehension over ``update_surrogate(module, trainer)``; 3 members by default, script.py:60), each on
its own bootstrapped batches.  On an MI355X one member's fused step is 64 workgroups (one per sequence)
on a 256-CU device and bound by the dependent-phase latency inside each workgroup, so the members do
not compete for anything: all members' steps are recorded as sibling branches of ONE hipGraph (each
member on its own stream forked from the capture stream, joined at the end) and the hardware runs them
side by side.

Constructions that do NOT work on ROCm 7 (measured, so nobody retries them): one graph per member
replayed on separate streams serialises; member graphs as child-graph nodes of a parent graph serialise
too (and lose the overlap inside a member); forked streams that fork again crash hipStreamEndCapture
(tools/dbg_capture.py) -- hence ``inner_forks(False)``: inside the ensemble graph a member keeps its own
step on one stream and the parallelism comes from the members.

Results are bit-identical to stepping the members in turn (same kernels, deterministic reductions).
"""
from typing import Sequence

import torch

from pdecontrol.surrogates import hipops
from pdecontrol.surrogates.graph_step import GraphedTBPTTStep, capture_graph


class EnsembleTBPTTStep:
    def __init__(self, modules: Sequence, batch_shape, action_shape=None, lr=None, warmup=3):
        """modules: PDETrainingModules on ONE CUDA device (distinct parameters); batch_shape: [B, T, 1, N]."""
        assert len(modules) > 0
        with hipops.inner_forks(False):
            self.members = [GraphedTBPTTStep(m, batch_shape, action_shape, lr=lr, warmup=warmup, capture=False)
                            for m in modules]
            self.device = self.members[0].device
            assert all(g.device == self.device for g in self.members), "ensemble members must share one GPU"
            # member 0 stays on the capture stream, the others are its siblings
            self.streams = hipops.pooled_streams(self.device, len(self.members) - 1, "member")
            self.graph = torch.cuda.CUDAGraph()
            self.stream = self.members[0].stream

            def record():
                cur = torch.cuda.current_stream(self.device)
                for st in self.streams:
                    st.wait_stream(cur)                      # fork
                for g, st in zip(self.members, [cur] + self.streams):
                    with torch.cuda.stream(st), g.capturing():
                        g.result = g._fwd_bwd()
                    if not g.adam_in_flush:
                        with torch.cuda.stream(st):
                            g.shared.opt.step()
                for st in self.streams:
                    cur.wait_stream(st)                      # join

            capture_graph(self.graph, record, self.stream)
            for g in self.members:
                g.logged = dict(g.module.__dict__.pop("_graph_logged", {}))

    def valid(self):
        """False once a member's captured launches are stale (parameters moved, delta statistics re-fitted between the
        controller's training rounds: ``GraphedTBPTTStep.valid``) -- build a new EnsembleTBPTTStep then; the Adam state
        belongs to the surrogates and carries over."""
        return all(g.valid() for g in self.members)

    def __len__(self):
        return len(self.members)

    def load(self, batches):
        """batches: one (states, actions) pair per member (bootstrapped batches differ per member)."""
        assert len(batches) == len(self.members)
        for g, (s, a) in zip(self.members, batches):
            g.states.copy_(s, non_blocking=True)
            g.actions.copy_(a, non_blocking=True)

    def step(self, batches=None):
        """One optimizer step of every member, concurrently.  Returns the members' static result dicts
        (overwritten by the next step)."""
        if batches is not None:
            self.load(batches)
        self.graph.replay()
        return [g.result for g in self.members]
