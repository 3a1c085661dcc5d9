"""Truncated-BPTT training module of the surrogate.

API mirror of the reference's ``pdecontrol/surrogates/training.py`` (``PDETrainingModule`` :14-62,
``training_step`` :64-130, ``validation_step`` :132-174, ``test_step`` :176-271,
``configure_optimizers`` :273-278): same constructor, same returned-dict keys, same logged metric
names ("Train Loss", "Val. Loss", ...).

What is different under the hood: logged values stay tensors (no ``.item()`` host sync inside
the step), time grids are Python-side, and ``fused_step`` offers the whole
forward + backward + Adam update as ONE replayable HIP graph for fixed batch shapes.

How the reference's caller reaches the fast paths (``pl.Trainer.fit``, pdecontrol/mbrl/mbrl.py:593):
  * default (automatic optimization): Lightning calls ``training_step`` eagerly, then ``backward`` and
    ``optimizer.step``.  On a CUDA device ``training_step`` runs on the fused HIP kernels (21 launches) and
    ``configure_optimizers`` returns a single-kernel (``fused=True``) Adam.
  * ``graphed=True`` (constructor keyword, i.e. ``--training '{"initial": {"graphed": true, ...}}'`` through the
    reference's own config channel, mbrl.py:230-245; or ``PDECONTROL_GRAPHED=1``): the module switches to Lightning's
    manual optimization and ``training_step`` replays the captured graph (forward + backward + Adam in one
    ``hipGraphLaunch``).  The optimizer returned by ``configure_optimizers`` then only carries the learning rate
    (schedulers keep working: the graph reads it from a device scalar); its ``step`` is never called.
"""
import os
from typing import Callable

import numpy as np
import torch

from pdecontrol._compat.lightning import pl
from pdecontrol.mbrl.types import ModelRollout
from pdecontrol.surrogates.surrogate import AutoRegPDESurrogate, LatentAutoRegPDESurrogate, PDESurrogate
from pdegym.common.transforms import BatchTransform, Identity, SampleTransform


class _GraphOwnedAdam(torch.optim.Adam):
    """What ``configure_optimizers`` hands to Lightning when the module is ``graphed``: it carries the parameter groups
    (learning rate, read by schedulers and copied into the captured step's device scalar) but its ``step`` applies
    nothing -- the Adam update already happened inside the replayed graph."""

    def step(self, closure=None):
        return None if closure is None else closure()


class PDETrainingModule(pl.LightningModule):
    def __init__(self, surrogate: PDESurrogate, loss: Callable, tstep: float, delta: float, env=None,
                 stransf: SampleTransform = None, undscaling: BatchTransform = None, tau: int = 5, tbtt: int = 10,
                 lr: float = 1e-03, lr_gamma: float = 1.0, step_size: int = 25, graphed: bool = None, **kwargs):
        super().__init__()
        self.graphed = (os.environ.get("PDECONTROL_GRAPHED", "0") == "1") if graphed is None else bool(graphed)
        # automatic optimization: forward and backward replayed as two captured graphs (PDECONTROL_SPLIT_GRAPHS=0 opts out)
        self.split_graphs = os.environ.get("PDECONTROL_SPLIT_GRAPHS", "1") != "0"

        if self.graphed:
            self.automatic_optimization = False   # Lightning: training_step owns backward + optimizer step
        self.surrogate, self.loss, self.tstep, self.delta, self.env = surrogate, loss, tstep, delta, env
        self.stransf = SampleTransform() if stransf is None else stransf
        self.undscaling = BatchTransform(Identity()) if undscaling is None else undscaling
        self.tau, self.tbtt, self.lr, self.lr_gamma, self.step_size = tau, tbtt, lr, lr_gamma, step_size
        if isinstance(surrogate, AutoRegPDESurrogate):
            self.training_mode = "delta"
        elif isinstance(surrogate, LatentAutoRegPDESurrogate):
            self.training_mode = "decoded"
        elif getattr(surrogate, "training_mode", None) in ("delta", "decoded"):
            self.training_mode = surrogate.training_mode      # e.g. the FNO surrogate of the Burgers path (f4)
        else:
            raise ValueError
        assert self.tbtt > self.tau, "Chunk size of TBTT must be larger than warm-up length."

    # -- helpers -----------------------------------------------------------------------------
    def _grid(self, n):
        """times / targets of an n-action chunk: k*tstep and (k+1)*tstep (CPU tensors)."""
        k = torch.arange(n)
        return self.tstep * k, self.tstep * (k + 1)

    def _full_rollout(self, states, actions) -> ModelRollout:
        times, targets = self._grid(actions.size(1))
        return self.surrogate.rollout(states=states[:, :self.tau], actions=actions, times=times, targets=targets,
                                      hidden=None)

    # -- training: truncated back-propagation through time ------------------------------------
    def tbptt_forward(self, states, actions):
        """Chunks of ``tbtt`` steps; chunk 0 warms up on the first ``tau`` true states, later chunks
        start from the previous chunk's last prediction with gradients cut (state and hidden)."""
        from pdecontrol.surrogates import ops
        if isinstance(self.surrogate, AutoRegPDESurrogate) and ops.use_fused_for(self.surrogate, states):
            # every chunk in a handful of launches, independent work on parallel streams (hipops._TBPTTFn)
            from pdecontrol.surrogates import hipops
            outputs, deltas, hidden, d_all = hipops.fused_tbptt(self.surrogate, states, actions, self.tau, self.tbtt)
            out = ModelRollout(outputs=outputs, deltas=deltas, hidden=tuple(h.detach() for h in hidden))
            out.time_major_deltas = d_all
            return [out]
        if states.is_cuda and ops.fused_enabled() and type(self.surrogate).__name__ == "FNOAutoRegSurrogate":
            # the FNO surrogate's whole TBPTT pass as one autograd node on the whole-network kernels (csrc/fno.hip); None:
            # a geometry they are not built for -> the generic chunk loop below (per-operator path)
            from pdecontrol.surrogates import fno_hip
            fused = fno_hip.tbptt(self.surrogate, states, actions, self.tau, self.tbtt, self._grid)
            if fused is not None:
                outputs, deltas, d_all = fused
                out = ModelRollout(outputs=outputs, deltas=deltas, hidden=())
                out.time_major_deltas = d_all
                return [out]
        rollouts = []
        seed_states, hidden = None, None
        autoreg = isinstance(self.surrogate, AutoRegPDESurrogate)
        if autoreg:
            self.surrogate.reencode_predictions = False  # inlatents are never read by the loss
        try:
            for c, achunk in enumerate(torch.split(actions, self.tbtt, dim=1)):
                if c == 0:
                    seed_states = states[:, :self.tbtt][:, :self.tau]
                times, targets = self._grid(achunk.size(1))
                out = self.surrogate.rollout(states=seed_states, actions=achunk, times=times, targets=targets,
                                             hidden=hidden)
                rollouts.append(out)
                seed_states = out.outputs[:, -1:].detach()
                out.hidden = hidden = tuple(h.detach() for h in out.hidden)
        finally:
            if autoreg:
                self.surrogate.reencode_predictions = True
        return rollouts

    def _fused_delta_loss(self, rollouts, states):
        """The loss section of training_step as one HIP launch, when the step ran on the fused kernels and the
        configuration is the controller's (delta mode, MSELoss(reduction="none"), affine undscaling)."""
        d_all = getattr(rollouts[0], "time_major_deltas", None) if len(rollouts) == 1 else None
        if d_all is None or self.training_mode != "delta":
            return None
        if not (isinstance(self.loss, torch.nn.MSELoss) and self.loss.reduction == "none"):
            return None
        from pdecontrol.surrogates import hipops
        consts = hipops.undscale_constants(self.undscaling)
        if consts is None or states.dtype != torch.float32 or states.shape[2] != 1:
            return None
        return hipops.fused_delta_loss(self.surrogate, d_all, states, self.delta, *consts)

    def _pipelined_training_step(self, batch):
        """The whole training pass -- forward, loss, backward, gradient reduction (+ Adam when the captured step lent its
        descriptors) -- in one hand-scheduled set of launches with the TBPTT chunks pipelined (``hipops.fused_tbptt_train``:
        chunk c's backward runs beside chunk c+1's forward).  No autograd and d loss = 1, so this is for the captured step
        only (``GraphedTBPTTStep``); returns None when the configuration is not the controller's (delta mode,
        MSELoss(reduction="none"), affine undscaling, fused kernels) or the sequence is a single chunk."""
        from pdecontrol.surrogates import ops
        states, actions, *_ = batch
        if not (isinstance(self.surrogate, AutoRegPDESurrogate) and ops.use_fused_for(self.surrogate, states)) or self.training_mode != "delta":
            return None
        if not (isinstance(self.loss, torch.nn.MSELoss) and self.loss.reduction == "none"):
            return None
        from pdecontrol.surrogates import hipops
        consts = hipops.undscale_constants(self.undscaling)
        if consts is None or states.dtype != torch.float32 or states.shape[2] != 1:
            return None
        with torch.no_grad():
            res = hipops.fused_tbptt_train(self.surrogate, states, actions, self.tau, self.tbtt, self.delta, *consts)
        if res is None:
            return None
        outputs, outdeltas, _hidden, loss, hsteploss, stats, deltas = res
        logged = {"Train Loss": loss, "Train Mean Delta Output": stats[0], "Train Std. Delta Output": stats[1],
                  "Train Mean Delta": stats[2], "Train Std. Delta": stats[3]}
        if torch.cuda.is_current_stream_capturing():
            self.__dict__["_graph_logged"] = logged
        else:
            for name, value in logged.items():
                self.log(name, value, on_step=False, on_epoch=True)
        return {"loss": loss, "hsteploss": hsteploss, "outputs": outputs, "actions": actions.detach(),
                "states": states.detach(), "outdeltas": outdeltas[:, :-1], "deltas": deltas}

    def training_step(self, batch, bidx):
        states = batch[0]
        if self.graphed and states.is_cuda:
            return self._graphed_training_step(batch)
        if self.split_graphs and states.is_cuda and torch.is_grad_enabled():
            out = self._split_graph_training_step(batch)
            if out is not None:
                return out
        return self._eager_training_step(batch, bidx)

    def _split_graph_training_step(self, batch):
        """training_step under Lightning's automatic optimization as two replayed hipGraphs behind one autograd node
        (``graph_step.GraphedAutogradStep``): forward + loss now, backward + gradient reduction when the caller runs
        ``loss.backward()``.  None when the configuration is not the controller's (then the launch-by-launch path runs)."""
        from pdecontrol.surrogates import ops
        states, actions, *_ = batch
        if not (isinstance(self.surrogate, AutoRegPDESurrogate) and ops.use_fused_for(self.surrogate, states)) or self.training_mode != "delta":
            return None
        if not (isinstance(self.loss, torch.nn.MSELoss) and self.loss.reduction == "none"):
            return None
        from pdecontrol.surrogates import hipops
        if hipops.undscale_constants(self.undscaling) is None or states.dtype != torch.float32 or states.shape[2] != 1:
            return None
        if torch.cuda.is_current_stream_capturing():
            return None
        from pdecontrol.surrogates.graph_step import GraphedAutogradStep
        key = (tuple(states.shape), tuple(actions.shape))
        cache = self.__dict__.setdefault("_split_steps", {})
        step = cache.get(key)
        if step is None or not step.valid():
            if not all(p.requires_grad for n, p in self.surrogate.named_parameters() if not n.endswith((".H0", ".C0"))):
                return None     # a frozen sub-module: the captured backward would still write its gradients
            step = cache[key] = GraphedAutogradStep(self, key[0], key[1])
        elif not step.all_trainable():
            return None
        out = step.forward(states, actions)
        for name, value in step.logged.items():
            self.log(name, value, on_step=False, on_epoch=True)
        return out

    def _eager_training_step(self, batch, bidx):
        states, actions, *_ = batch
        rollouts = self.tbptt_forward(states, actions)

        fused = self._fused_delta_loss(rollouts, states)
        if fused is not None:
            # loss, per-step loss, the four logged statistics and d loss / d deltas: one launch
            loss, hsteploss, stats, deltas = fused
            outputs, outdeltas = rollouts[0].outputs, rollouts[0].deltas[:, :-1]
            m_out, s_out, m_true, s_true = stats[0], stats[1], stats[2], stats[3]
        else:
            outputs = torch.cat([r.outputs for r in rollouts], dim=1)
            outdeltas = torch.cat([r.deltas for r in rollouts], dim=1)[:, :-1]
            deltas = self.undscaling(torch.diff(states, dim=1) / self.delta)
            if self.training_mode == "delta":
                loss = self.loss(outdeltas, deltas)
            else:
                decoded = torch.cat((states[:, :1], outputs[:, :-1]), dim=1)
                loss = self.loss(decoded, states)
            hsteploss = loss.mean(dim=(0, 2, 3))
            loss = loss.mean()
            m_out, s_out = outdeltas.detach().mean(), outdeltas.detach().std()
            m_true, s_true = deltas.detach().mean(), deltas.detach().std()

        logged = {"Train Loss": loss.detach(), "Train Mean Delta Output": m_out.detach(), "Train Std. Delta Output": s_out.detach(),
                  "Train Mean Delta": m_true.detach(), "Train Std. Delta": s_true.detach()}
        if torch.cuda.is_available() and states.is_cuda and torch.cuda.is_current_stream_capturing():
            self.__dict__["_graph_logged"] = logged     # static tensors of the captured step, logged at every replay
        else:
            for name, value in logged.items():
                self.log(name, value, on_step=False, on_epoch=True)

        return {"loss": loss, "hsteploss": hsteploss.detach(), "outputs": outputs.detach(),
                "actions": actions.detach(), "states": states.detach(), "outdeltas": outdeltas.detach(),
                "deltas": deltas.detach()}

    def fused_step(self, batch, lr=None):
        """One optimizer step -- training_step + backward + Adam(lr) -- as ONE replayed HIP graph on static
        buffers (pdecontrol.surrogates.graph_step.GraphedTBPTTStep; a graph per batch shape, captured on first
        use; ONE Adam state and learning rate for all of them).  Returns training_step's dict; its tensors are
        overwritten by the next call.  CUDA only."""
        from pdecontrol.surrogates.graph_step import GraphedTBPTTStep
        states, actions, *_ = batch
        key = (tuple(states.shape), tuple(actions.shape))
        cache = self.__dict__.setdefault("_graphed_steps", {})
        if key not in cache or not cache[key].valid():
            # one process per GPU with an initialised process group: the captured step must exchange gradients
            # (forward/backward graph -> one flat-bucket all-reduce -> Adam graph), never train the ranks apart silently
            dist = torch.distributed
            distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
            cache[key] = GraphedTBPTTStep(self, key[0], key[1], lr=lr, distributed=distributed)
        self.__dict__["_last_graphed_step"] = cache[key]
        return cache[key].step(states, actions, lr=lr)

    def _graphed_training_step(self, batch):
        """training_step under Lightning's manual optimization: replay the graph with the learning rate of the
        optimizer Lightning holds (so StepLR and friends act on the captured Adam), log what training_step logs."""
        lr = None
        try:
            opts = self.optimizers()
            opt = opts[0] if isinstance(opts, (list, tuple)) else opts
            lr = float(opt.param_groups[0]["lr"])
        except Exception:   # no trainer attached (direct call): the module's own lr
            lr = None
        out = self.fused_step(batch, lr=lr)
        if lr is not None:
            opt.step()      # bookkeeping only (_GraphOwnedAdam): keeps scheduler / Lightning step counters consistent
        # the five "Train ..." metrics of the step that was just replayed (static tensors of that captured step)
        for name, value in self.__dict__["_last_graphed_step"].logged.items():
            self.log(name, value, on_step=False, on_epoch=True)
        return out

    def on_train_epoch_end(self):
        # manual optimization: Lightning does not step the schedulers for us
        if self.graphed:
            try:
                scheds = self.lr_schedulers()
            except Exception:
                return
            if scheds is None:
                return
            for sch in (scheds if isinstance(scheds, (list, tuple)) else [scheds]):
                sch.step()

    # -- validation / test: one un-truncated rollout ------------------------------------------
    def validation_step(self, batch, bidx):
        states, actions, *_ = batch
        out = self._full_rollout(states, actions)
        decoded = torch.cat((states[:, :1], out.outputs[:, :-1]), dim=1)  # IC-augmented prediction
        outdeltas = out.deltas[:, :-1]
        deltas = self.undscaling(torch.diff(states, dim=1) / self.delta)
        self.log("Val. Delta Loss", self.loss(outdeltas, deltas).detach().mean(), on_step=False, on_epoch=True)
        self.log("Val. Scaled Loss", self.loss(decoded, states).mean(), on_step=False, on_epoch=True)

        states = self.stransf.otransf.Inverse(states)
        decoded = self.stransf.otransf.Inverse(decoded)
        loss = self.loss(decoded, states)
        hsteploss = loss.detach().mean(dim=(0, 2, 3))
        loss = loss.mean()
        self.log("Val. Loss", loss, on_step=False, on_epoch=True)
        return {"loss": loss.detach(), "hsteploss": hsteploss.detach(), "outputs": decoded.detach(),
                "actions": actions.detach(), "states": states.detach(), "outdeltas": outdeltas.detach(),
                "deltas": deltas.detach()}

    def test_step(self, batch, bidx):
        states, actions, *_ = batch
        out = self._full_rollout(states, actions)
        outputs = torch.cat((states[:, :1], out.outputs[:, :-1]), dim=1)
        states = self.stransf.otransf.Inverse(states).detach().cpu()
        outputs = self.stransf.otransf.Inverse(outputs).detach().cpu()
        actions = actions.detach().cpu()

        def norms(a, b, dims, p):
            return torch.norm(a - b, p=p, dim=dims[0]), torch.norm(b, p=p, dim=dims[0])

        err1, ref1 = norms(outputs, states, (3,), 1)
        err2, ref2 = norms(outputs, states, (3,), 2)
        tm = lambda v: v.mean(dim=(0, 2)).numpy()
        data = {"MSE": self.loss(outputs, states).mean().numpy(), "l1_loss": tm(err1), "l2_loss": tm(err2),
                "l1_loss_scaled": tm(err1 / ref1), "l2_loss_scaled": tm(err2 / ref2),
                "nrmse": tm(err2 ** 2 / ref2 ** 2)}

        # reward estimates on true vs predicted states (env.forcing / env.reward_func, per sample)
        b, t, ac, ah = actions.shape
        _, _, sc, sh = states.shape
        flat = lambda v, c, h: v.reshape(b * t, c, h)
        phi = BatchTransform(self.env.forcing)(self.stransf.atransf.Inverse(flat(actions, ac, ah)))
        rew = lambda vals: torch.stack([torch.as_tensor(self.env.reward_func(v, p)) for v, p in zip(vals, phi)],
                                       dim=0).reshape(b, t)
        rews, pred_rews = rew(flat(states, sc, sh)), rew(flat(outputs, sc, sh))
        e1, r1 = torch.norm(rews - pred_rews, p=1, dim=0), torch.norm(rews, p=1, dim=0)
        e2, r2 = torch.norm(rews - pred_rews, p=2, dim=0), torch.norm(rews, p=2, dim=0)
        data.update({"l1_loss_rews": e1.numpy(), "l2_loss_rews": e2.numpy(), "l1_loss_scaled_rews": (e1 / r1).numpy(),
                     "l2_loss_scaled_rews": (e2 / r2).numpy(), "nrmse_rews": (e2 ** 2 / r2 ** 2).numpy()})

        # spatial derivatives (env.rhs) of true vs predicted states.  The reference calls env.rhs once per sample
        # (training.py:237-245); the HIP rhs hook is row-parallel, so all B*T samples go in ONE call (same values)
        def derivatives(vals):
            _, (ux, uxx, uxxxx) = self.env.rhs(vals.numpy(), phi.numpy())
            d = torch.as_tensor(np.stack([ux, uxx, uxxxx], axis=1))            # [B*T, 3, C, H]
            return d.reshape(b, t, *d.shape[1:])
        dv, pdv = derivatives(flat(states, sc, sh)), derivatives(flat(outputs, sc, sh))
        d1, n1 = torch.norm(dv - pdv, p=1, dim=4), torch.norm(dv, p=1, dim=4)
        d2, n2 = torch.norm(dv - pdv, p=2, dim=4), torch.norm(dv, p=2, dim=4)
        dm = lambda v: v.mean(dim=(0, 3))
        named = {"l1_loss_derivs": dm(d1), "l2_loss_derivs": dm(d2), "l1_loss_scaled_derivs": dm(d1 / n1),
                 "l2_loss_scaled_derivs": dm(d2 / n2), "nrms_derivs": dm(d2 ** 2 / n2 ** 2)}
        for name, table in named.items():
            for idx, column in enumerate(table.T):
                data[f"{name}-derivative-{idx}"] = column
        data.update({"states": states.numpy(), "outputs": outputs.numpy(), "actions": actions.numpy()})
        return data

    def _fused_eager(self):
        from pdecontrol.surrogates import hipops, ops
        return ops.fused_enabled() and hipops.fused_supported(self.surrogate)

    def configure_optimizers(self):
        params = list(self.surrogate.parameters())
        # same update rule as the reference's Adam; on a GPU as one multi-tensor kernel instead of ~6 per parameter
        on_gpu = bool(params) and params[0].is_cuda
        if self.graphed and on_gpu:
            optimizer = _GraphOwnedAdam(params, lr=self.lr)
        elif on_gpu and self._fused_eager():
            # the whole Adam update as ONE launch over the fused packs' flat state (same rule as torch.optim.Adam)
            from pdecontrol.surrogates import hipops
            optimizer = hipops.PackAdam(self.surrogate, lr=self.lr)
        else:
            extra = {"fused": True} if on_gpu else {}
            optimizer = torch.optim.Adam(params, lr=self.lr, **extra)
        scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=self.step_size, gamma=self.lr_gamma)
        return [optimizer], [{"scheduler": scheduler, "interval": "epoch"}]
