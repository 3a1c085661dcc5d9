"""One TBPTT optimizer step as a replayable HIP graph.

At the reference's batch size (B = 64, 9.7 k parameters) the step is ~4 000 tiny kernels: on a GPU
it is bound by launch latency, not by arithmetic.  MI355X-native answer: capture forward + backward
(+ Adam) once into a hipGraph on static buffers and replay it every step -- no Python, no per-kernel
launch cost, no host synchronisation inside the step.

Single GPU : [ zero grads | forward | backward | Adam ]                       = one graph
             (on the fused kernels: [ forward | backward ] only -- the launches that reduce the partial
             gradient rows apply the Adam update themselves, and nothing needs zeroing)
Data parallel: [ zero | forward | backward ] -> all-reduce(flat bucket) -> [ Adam ]  = two graphs with one
             RCCL call between them (pdecontrol.surrogates.distributed.FlatGradBucket).
"""
import torch

from pdecontrol.surrogates.distributed import FlatGradBucket


class GraphedTBPTTStep:
    """``step()`` replays one optimizer step.  On the fused kernels (single GPU) the Adam update is part of the captured
    gradient-reduction launches: ``self.opt`` then only documents the hyper-parameters, its state is not advanced."""

    def __init__(self, module, batch_shape, action_shape=None, lr=None, distributed=False, warmup=3, capture=True):
        """module: PDETrainingModule on a CUDA device; batch_shape: [B, T, 1, N] of states.
        capture=False prepares everything (static buffers, warmed-up kernels, fresh Adam state) but leaves
        the capture to the caller (EnsembleTBPTTStep records several members into one graph)."""
        self.module = module
        dev = next(module.surrogate.parameters()).device
        assert dev.type == "cuda", "HIP graphs need the module on a GPU"
        self.device = dev
        # static buffers: time-major storage behind the module's [B, T, 1, N] view, so that the time-major copies the
        # fused path wants (hipops._TBPTTFn) are no-ops instead of two transposing kernels per step
        tm = lambda shape: torch.zeros((shape[1], shape[0]) + tuple(shape[2:]), device=dev).transpose(0, 1)
        self.states = tm(tuple(batch_shape))
        self.actions = tm(tuple(action_shape or batch_shape))
        self.distributed = distributed
        self.bucket = FlatGradBucket(module.surrogate.parameters())
        self.lr = lr if lr is not None else module.lr
        self.opt = self._make_adam()
        self.result = None
        self.g_main = self.g_opt = None
        self.adam_in_flush = False
        self._prepare(warmup)
        if capture:
            self._capture()

    def _make_adam(self):
        """Adam as ONE multi-tensor kernel (fused=True) instead of ~3 tiny kernels per parameter."""
        params = list(self.module.surrogate.parameters())
        try:
            return torch.optim.Adam(params, lr=self.lr, capturable=True, fused=True)
        except (RuntimeError, TypeError, ValueError):
            return torch.optim.Adam(params, lr=self.lr, capturable=True)

    def _fwd_bwd(self):
        if not self.adam_in_flush:
            self.bucket.zero_()
        out = self.module.training_step((self.states, self.actions), 0)
        from pdecontrol.surrogates import hipops, ops
        if ops.fused_enabled():
            out["loss"].backward(gradient=hipops.unit_grad(self.device))
        else:
            out["loss"].backward()
        return out

    def _prepare(self, warmup):
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        # warm-up on a side stream with the real optimizer state untouched: snapshot and restore
        snap = [p.detach().clone() for p in self.module.surrogate.parameters()]
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._fwd_bwd()
                self.opt.step()
        torch.cuda.current_stream(self.device).wait_stream(side)
        with torch.no_grad():
            for p, s in zip(self.module.surrogate.parameters(), snap):
                p.copy_(s)
        self.opt = self._make_adam()
        # Adam state must exist before capture: one throw-away step on zero grads, then reset
        self.bucket.zero_()
        self.opt.step()
        with torch.no_grad():
            for p, s in zip(self.module.surrogate.parameters(), snap):
                p.copy_(s)
            for st in self.opt.state.values():
                st["step"].zero_()
                st["exp_avg"].zero_()
                st["exp_avg_sq"].zero_()
        # fused kernels, single GPU: the flush launches take the Adam step (fresh state, like self.opt's)
        # (the descriptors -- moment buffers, step counter -- are created here and handed to the packs only while
        #  the graph is being captured: the captured launches carry them by value, and any other backward pass
        #  through the same surrogate keeps accumulating plain gradients)
        packs = getattr(self.module.surrogate, "_fused_packs", None)
        from pdecontrol.surrogates import ops
        if packs is not None and ops.fused_enabled() and not self.distributed:
            betas, eps = self.opt.defaults["betas"], self.opt.defaults["eps"]
            packs.enable_adam(self.lr, betas, eps)
            self._adam_state = [pack.adam for pack in packs.packs]
            packs.disable_adam()
            self.adam_in_flush = True
        torch.cuda.synchronize(self.device)

    def capturing(self):
        """Context manager for the duration of a hipGraph capture of ``_fwd_bwd``: lends the Adam descriptors to the
        surrogate's packs (no-op unless the optimizer step lives in the flush launches)."""
        step = self

        class _Lend:
            def __enter__(self_inner):
                if step.adam_in_flush:
                    for pack, state in zip(step.module.surrogate._fused_packs.packs, step._adam_state):
                        pack.adam = state

            def __exit__(self_inner, *exc):
                if step.adam_in_flush:
                    step.module.surrogate._fused_packs.disable_adam()

        return _Lend()

    def _capture(self):
        self.g_main = torch.cuda.CUDAGraph()
        if not self.distributed:
            with self.capturing(), torch.cuda.graph(self.g_main):
                self.result = self._fwd_bwd()
                if not self.adam_in_flush:
                    self.opt.step()
            self.g_opt = None
        else:
            with torch.cuda.graph(self.g_main):
                self.result = self._fwd_bwd()
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt):
                self.opt.step()

    def step(self, states=None, actions=None):
        """Copy the batch into the static buffers (if given) and replay.  Returns the static
        result dict of training_step (tensors are overwritten by the next replay)."""
        if states is not None:
            self.states.copy_(states, non_blocking=True)
        if actions is not None:
            self.actions.copy_(actions, non_blocking=True)
        self.g_main.replay()
        if self.g_opt is not None:
            self.bucket.all_reduce_mean()
            self.g_opt.replay()
        return self.result
