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

What is shared and what is per batch shape: a graph is tied to its static input buffers, so there is one
``GraphedTBPTTStep`` per (states, actions) shape; the OPTIMIZER is tied to the parameters, so every step object of the
same module uses the same Adam moments / step counter / learning rate (``_SharedTrainState``) -- a ragged last batch or
a curriculum change of T continues the same optimisation, like the reference's single ``torch.optim.Adam``
(pdecontrol/surrogates/training.py:273-278).

Capture hygiene (the round-1 SIGABRT): torch >= 2.9 no longer runs ``gc.collect()`` when a capture starts.  If
Python's cyclic collector fires INSIDE the capture and frees GPU objects of earlier work (graphs, events, tensors with
recorded stream uses), their destructors issue HIP calls that are illegal while a stream is capturing and the
process aborts.  ``capture_graph`` therefore collects before the capture and keeps the collector off during it;
warm-up and capture run on the same explicit stream, and only detached tensors are kept from the captured step.
"""
import gc
import os

import torch

from pdecontrol.surrogates.distributed import FlatGradBucket


def capture_graph(graph, fn, stream):
    """Record ``fn()`` into ``graph`` on ``stream`` with the cyclic garbage collector parked (see module docstring)."""
    gc.collect()
    torch.cuda.synchronize(stream.device)
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, stream=stream):
            out = fn()
    finally:
        if was_enabled:
            gc.enable()
    return out


def _detached(result):
    return {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in result.items()}


class _SharedTrainState:
    """Per-module state every captured step shares: the flat gradient bucket (param.grad are views of it), the
    warm-up / capture stream, the torch Adam of the paths that do not fold the update into the flush launch, and the
    learning rate."""

    def __init__(self, module, lr):
        params = list(module.surrogate.parameters())
        self.device = params[0].device
        self.bucket = FlatGradBucket(params)
        from pdecontrol.surrogates.hipops import pooled_streams
        (self.stream,) = pooled_streams(self.device, 1, "capture")
        self.lr = float(lr)
        self._params = params
        self.opt = None

    def torch_adam(self):
        """Adam as ONE multi-tensor kernel, capturable, learning rate in a device scalar; state created (zeroed) here so
        that a capture never allocates it."""
        if self.opt is not None:
            return self.opt
        lr = torch.tensor(self.lr, device=self.device, dtype=torch.float32)
        try:
            opt = torch.optim.Adam(self._params, lr=lr, capturable=True, fused=True)
        except (RuntimeError, TypeError, ValueError):
            opt = torch.optim.Adam(self._params, lr=lr, capturable=True)
        snap = [p.detach().clone() for p in self._params]
        self.bucket.zero_()
        opt.step()
        with torch.no_grad():
            for p, s in zip(self._params, snap):
                p.copy_(s)
            for st in opt.state.values():
                st["step"].zero_()
                st["exp_avg"].zero_()
                st["exp_avg_sq"].zero_()
        self.opt = opt
        return opt

    def set_lr(self, lr, packs=None):
        lr = float(lr)
        if packs is not None:
            packs.set_lr(lr)
        if lr != self.lr and self.opt is not None:
            for group in self.opt.param_groups:
                group["lr"].fill_(lr)
        self.lr = lr


def shared_state(module, lr=None):
    state = module.__dict__.get("_graph_state")
    first = next(module.surrogate.parameters())
    if state is None or state.device != first.device or state._params[0] is not first:
        state = _SharedTrainState(module, module.lr if lr is None else lr)
        module.__dict__["_graph_state"] = state
    elif lr is not None:
        state.set_lr(lr, getattr(module.surrogate, "_fused_packs", None))
    return state


class GraphedTBPTTStep:
    """``step()`` replays one optimizer step of ``module`` for one batch shape."""

    def __init__(self, module, batch_shape, action_shape=None, lr=None, distributed=False, warmup=3, capture=True,
                 pipelined=None):
        """module: PDETrainingModule on a CUDA device; batch_shape: [B, T, 1, N] of states.
        capture=False prepares everything (static buffers, warmed-up kernels, optimizer state) but leaves
        the capture to the caller (EnsembleTBPTTStep records several members into one graph)."""
        self.module = module
        # pipelined: chunk c's backward beside chunk c+1's forward (hipops.fused_tbptt_train).  True / False force one schedule;
        # None (default, PDECONTROL_PIPELINED=auto) captures both and keeps the faster one: the pipelined graph needs its two
        # node lists on two different hardware queues, and which queue ROCm gives the graph's second stream depends on how
        # many streams the process has created (4 hardware queues): on an unlucky mapping the branches serialise and the
        # combined schedule wins.  PDECONTROL_PIPELINED=0 / 1 force it from the environment.
        env = os.environ.get("PDECONTROL_PIPELINED", "auto")
        if pipelined is None and env in ("0", "1"):
            pipelined = env == "1"
        self.autotune = pipelined is None
        self.pipelined = True if pipelined is None else bool(pipelined)
        self.used_pipelined = False
        dev = next(module.surrogate.parameters()).device
        assert dev.type == "cuda", "HIP graphs need the module on a GPU"
        self.device = dev
        # static buffers: time-major storage behind the module's [B, T, 1, N] view, so that the time-major copies the
        # fused path wants (hipops._TBPTTFn) are no-ops instead of two transposing kernels per step
        tm = lambda shape: torch.zeros((shape[1], shape[0]) + tuple(shape[2:]), device=dev).transpose(0, 1)
        self.states = tm(tuple(batch_shape))
        self.actions = tm(tuple(action_shape or batch_shape))
        self.distributed = distributed
        self.shared = shared_state(module, lr)
        self.bucket = self.shared.bucket
        self.stream = self.shared.stream
        self.result = None
        self.g_main = self.g_opt = None
        self.adam_in_flush = False
        self._adam_state = None
        self.logged = {}
        from pdecontrol.surrogates import hipops
        self._scaling = hipops.scaling_signature(module.surrogate, getattr(module, "undscaling", None))
        self._ptr0 = next(module.surrogate.parameters()).data_ptr()
        self._prepare(warmup)
        if capture:
            self._capture()

    def valid(self):
        """False once what the captured launches carry by value has changed: the parameters' addresses, or the delta
        scaling statistics (re-fitted between the controller's training rounds, mbrl.py:597-602)."""
        from pdecontrol.surrogates import hipops
        params = self.shared._params
        first = next(self.module.surrogate.parameters())
        if not params or first is not params[0] or first.data_ptr() != self._ptr0:
            return False
        return hipops.same_signature(self._scaling, hipops.scaling_signature(self.module.surrogate,
                                                                             getattr(self.module, "undscaling", None)))

    @property
    def lr(self):
        return self.shared.lr

    @property
    def opt(self):
        """The torch optimizer of the paths that need one (plain torch kernels, data-parallel); None when the Adam update
        lives in the captured flush launches."""
        return None if self.adam_in_flush else self.shared.torch_adam()

    def _fused(self):
        from pdecontrol.surrogates import ops
        return ops.use_fused_for(self.module.surrogate, self.states)

    def _fwd_bwd(self):
        if not self.adam_in_flush:
            self.bucket.zero_()
        if self.pipelined and self._fused():
            # forward, loss, backward and gradient reduction hand-scheduled with the TBPTT chunks pipelined (no autograd)
            out = self.module._pipelined_training_step((self.states, self.actions))
            self.used_pipelined = out is not None
            if out is not None:
                return _detached(out)
        out = self.module._eager_training_step((self.states, self.actions), 0)
        if self._fused():
            from pdecontrol.surrogates import hipops
            out["loss"].backward(gradient=hipops.unit_grad(self.device))
        else:
            out["loss"].backward()
        return _detached(out)

    def _prepare(self, warmup):
        """Warm the kernels / allocator on the capture stream.  Parameters and optimizer state are left untouched:
        warm-up passes only compute gradients."""
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                self._fwd_bwd()
            if self.autotune and self.used_pipelined:
                # the schedule autotune (_capture) also captures the combined (autograd) schedule: it gets its eager
                # passes here too, so that nothing persistent -- the unit root gradient, the packs' partial-gradient rows
                # at the size THAT schedule needs, the autograd graph's buffers -- is first allocated under capture
                self.pipelined = False
                try:
                    for _ in range(max(1, min(warmup, 2))):
                        self._fwd_bwd()
                finally:
                    self.pipelined = self.used_pipelined = True
            packs = getattr(self.module.surrogate, "_fused_packs", None)
            if packs is not None and self._fused() and not self.distributed:
                # fused kernels, single GPU: the flush launches take the Adam step.  The descriptors (the surrogate's
                # ONE set of moments / step counter / device learning rate) are lent to the packs only while a graph is
                # being captured: the captured launches carry them by value, and any other backward pass through the
                # same surrogate keeps accumulating plain gradients.
                betas, eps = (0.9, 0.999), 1e-8
                self._adam_state = packs.adam_descriptors(self.shared.lr, betas, eps)
                self.adam_in_flush = True
            else:
                self.shared.torch_adam()
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        torch.cuda.synchronize(self.device)

    def capturing(self):
        """Context manager for the duration of a hipGraph capture of ``_fwd_bwd``: lends the Adam descriptors to the
        surrogate's packs (no-op unless the optimizer step lives in the flush launches)."""
        return _LendAdam(self)

    def _capture(self):
        self._capture_main()
        if self.autotune and self.used_pipelined:
            t_pipe = self._time_replays()
            keep = (self.g_main, self.result, self.logged)
            self.pipelined = False
            self._capture_main()
            t_comb = self._time_replays()
            self.schedule_times_ms = {"pipelined": t_pipe, "combined": t_comb}
            if t_pipe <= t_comb:
                self.g_main, self.result, self.logged = keep
                self.pipelined = self.used_pipelined = True
        if self.distributed:
            self.g_opt = torch.cuda.CUDAGraph()
            capture_graph(self.g_opt, self.shared.opt.step, self.stream)

    def _time_replays(self, warm=3, reps=10):
        """Mean replay time of ``g_main`` in ms, with the parameters and the Adam state put back afterwards."""
        packs = getattr(self.module.surrogate, "_fused_packs", None)
        tensors = list(self.shared._params)
        if packs is not None:
            for pack in packs.packs:
                if pack._adam_state is not None:
                    tensors.extend(pack._adam_state)
        if self.shared.opt is not None:
            for st in self.shared.opt.state.values():
                tensors.extend(t for t in st.values() if isinstance(t, torch.Tensor))
        snap = [t.detach().clone() for t in tensors]
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(warm):
            self.g_main.replay()
        start.record()
        for _ in range(reps):
            self.g_main.replay()
        end.record()
        torch.cuda.synchronize(self.device)
        with torch.no_grad():
            for t, sv in zip(tensors, snap):
                t.copy_(sv)
        torch.cuda.synchronize(self.device)
        return start.elapsed_time(end) / reps

    def _capture_main(self):
        self.g_main = torch.cuda.CUDAGraph()

        def main():
            with self.capturing():
                res = self._fwd_bwd()
            if not self.adam_in_flush and not self.distributed:
                self.shared.opt.step()
            return res

        self.result = capture_graph(self.g_main, main, self.stream)
        # training_step's logged metrics of THIS captured step (static tensors, refreshed by every replay)
        self.logged = dict(self.module.__dict__.pop("_graph_logged", {}))

    def set_lr(self, lr):
        self.shared.set_lr(lr, getattr(self.module.surrogate, "_fused_packs", None))

    def step(self, states=None, actions=None, lr=None):
        """Copy the batch into the static buffers (if given) and replay.  Returns the static
        result dict of training_step (detached tensors, overwritten by the next replay)."""
        if lr is not None:
            self.set_lr(lr)
        if states is not None:
            self.states.copy_(states, non_blocking=True)
        if actions is not None:
            self.actions.copy_(actions, non_blocking=True)
        self.g_main.replay()
        if self.g_opt is not None:
            self.bucket.all_reduce_mean()
            self.g_opt.replay()
        return self.result


class _LendAdam:
    def __init__(self, step):
        self.step = step

    def __enter__(self):
        if self.step.adam_in_flush:
            self.step.module.surrogate._fused_packs.lend_adam(self.step._adam_state)

    def __exit__(self, *exc):
        if self.step.adam_in_flush:
            self.step.module.surrogate._fused_packs.disable_adam()


class _ReplayBackward(torch.autograd.Function):
    """The autograd node behind the loss a ``GraphedAutogradStep`` returns: its backward replays the captured backward."""

    @staticmethod
    def forward(ctx, anchor, loss_static, step):
        ctx.step = step
        return loss_static.clone()

    @staticmethod
    def backward(ctx, g):
        ctx.step.backward(g)
        return None, None, None


class GraphedAutogradStep:
    """``training_step`` for Lightning's AUTOMATIC optimization (``pl.Trainer.fit``'s default, pdecontrol/mbrl/mbrl.py:593) as
    two replayed hipGraphs behind one autograd node, for one batch shape:

      forward()            copies the batch into static buffers, replays  [ TBPTT forward | delta loss ]  and returns
                           training_step's dict; ``loss`` carries a grad_fn
      loss.backward()      replays  [ every chunk's backward | encoder backward | gradient reduction ]  into the packs' flat
                           gradient buffers and re-attaches their views to ``param.grad``
      optimizer.step()     whatever Lightning holds (``hipops.PackAdam``: one launch)

    Lightning's closure order (training_step -> zero_grad -> backward -> step), gradient clipping, accumulation
    (``param.grad`` already defined: the captured "add" variant, or views + add for foreign tensors) and loss scaling
    (the incoming gradient multiplies d loss / d deltas before the replay) keep their meaning; the ~0.8 ms of Python that the
    launch-by-launch eager step costs per batch does not.  The chunks are not pipelined here (backward starts when Lightning
    says so); ``GraphedTBPTTStep`` (``graphed=True``) is the fastest route."""

    def __init__(self, module, batch_shape, action_shape=None, warmup=2):
        from pdecontrol.surrogates import hipops
        self.module = module
        dev = next(module.surrogate.parameters()).device
        assert dev.type == "cuda"
        self.device = dev
        tm = lambda shape: torch.zeros((shape[1], shape[0]) + tuple(shape[2:]), device=dev).transpose(0, 1)
        self.states = tm(tuple(batch_shape))
        self.actions = tm(tuple(action_shape or batch_shape))
        (self.stream,) = hipops.pooled_streams(dev, 1, "capture")
        consts = hipops.undscale_constants(module.undscaling)
        args = (module.surrogate, self.states, self.actions, module.tau, module.tbtt, module.delta) + tuple(consts)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream), torch.no_grad():
            for _ in range(warmup):                   # kernels, allocator, partial-row buffers: nothing may grow under capture
                st, _res = hipops.fused_tbptt_forward_loss(*args)
                hipops.fused_tbptt_backward(st, accumulate=False)
        torch.cuda.current_stream(dev).wait_stream(self.stream)
        torch.cuda.synchronize(dev)
        self.g_fwd, self.g_bwd, self.g_bwd_acc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        pool = torch.cuda.graph_pool_handle()

        def fwd():
            with torch.no_grad():
                return hipops.fused_tbptt_forward_loss(*args)

        self._st, res = self._capture(self.g_fwd, fwd, pool)
        # the backward graphs read d loss / d deltas times the incoming gradient from a buffer of their own: the forward
        # graph's output stays intact, so backward() may run more than once per forward (retain_graph)
        self._st.dd_in = torch.zeros_like(self._st.dd_all)
        self._capture(self.g_bwd, lambda: hipops.fused_tbptt_backward(self._st, accumulate=False), pool)
        self._capture(self.g_bwd_acc, lambda: hipops.fused_tbptt_backward(self._st, accumulate=True), pool)
        outputs, outdeltas, _hidden, loss, hsteploss, stats, deltas = res
        self.loss = loss
        self.result = {"hsteploss": hsteploss, "outputs": outputs, "actions": self.actions, "states": self.states,
                       "outdeltas": outdeltas[:, :-1], "deltas": deltas}
        self.logged = {"Train Loss": loss, "Train Mean Delta Output": stats[0], "Train Std. Delta Output": stats[1],
                       "Train Mean Delta": stats[2], "Train Std. Delta": stats[3]}
        self.packs = module.surrogate._fused_packs
        self._key = self.packs.key
        self._scaling = hipops.scaling_signature(module.surrogate, module.undscaling)

    def _capture(self, graph, fn, pool):
        gc.collect()
        torch.cuda.synchronize(self.device)
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(graph, pool=pool, stream=self.stream):
                out = fn()
        finally:
            if was_enabled:
                gc.enable()
        return out

    def valid(self):
        """False once the surrogate's parameters have been re-allocated or the delta scaling statistics re-fitted (the graphs
        carry addresses and constants by value)."""
        from pdecontrol.surrogates import hipops
        packs = getattr(self.module.surrogate, "_fused_packs", None)
        if packs is not self.packs or packs.key != type(packs)._key(self.module.surrogate, packs.n):
            return False
        return hipops.same_signature(self._scaling, hipops.scaling_signature(self.module.surrogate, self.module.undscaling))

    def all_trainable(self):
        return all(p.requires_grad for pack in self.packs.packs for p in pack.params)

    def forward(self, states, actions):
        self.states.copy_(states, non_blocking=True)
        self.actions.copy_(actions, non_blocking=True)
        self.g_fwd.replay()
        out = dict(self.result)
        out["loss"] = _ReplayBackward.apply(self.packs.anchor, self.loss, self)
        return out

    def backward(self, g):
        packs = self.packs.packs
        torch.mul(self._st.dd_all, g, out=self._st.dd_in)   # d loss / d deltas of the captured loss times the incoming gradient
        state = []
        for pack in packs:
            grads = [p.grad for p in pack.params]
            if all(x is None for x in grads):
                state.append("none")
            elif pack.grads_are_own_views():
                state.append("own")
            else:
                state.append("foreign")
        if all(s == "own" for s in state):
            self.g_bwd_acc.replay()      # accumulation onto what the previous backward left in the flat buffers
            return
        keep = [pack.gflat.clone() if s == "own" else None for pack, s in zip(packs, state)]
        self.g_bwd.replay()
        for pack, s, k in zip(packs, state, keep):
            if s == "none":
                for p, view in zip(pack.params, pack._gviews):
                    p.grad = view
            elif s == "own":
                pack.gflat.add_(k)
            else:
                for p, view in zip(pack.params, pack._gviews):
                    if p.grad is None:
                        p.grad = view.clone()
                    else:
                        p.grad.add_(view)
