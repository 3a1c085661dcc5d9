"""ctypes binding + autograd glue of libsurrogate_hip.so (C ABI: include/surrogate_hip.h).

``fused_rollout`` is the GPU implementation of ``AutoRegPDESurrogate.rollout`` for the
``KSAutoRegConvolutionalLSTM`` family: two encoder launches (all given states, all actions) and ONE
launch for the whole chunk of time steps (ConvLSTM cell + decoder + integration per step, time loop
inside the kernel), each with a matching backward launch.  Parameter gradients never pass through
autograd: the backward kernels add them to per-workgroup rows of a partial buffer and a flush
kernel -- queued to run when the backward pass ends -- reduces the rows into ``param.grad``.  The
autograd graph only carries activations; a zero-dim ``anchor`` tensor makes the custom Functions
differentiable even when their data inputs are not.

There is no fallback in here: if the library is missing, ``load()`` raises.
"""
import ctypes
import os

import types

import torch
from torch import nn

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.abspath(os.path.join(_HERE, "..", "..", "lib", "libsurrogate_hip.so"))

RB_NPARAM = 9
ST_NPARAM = 26
_fp = ctypes.c_void_p


class EncoderParams(ctypes.Structure):
    _fields_ = [("w", _fp * (3 * RB_NPARAM)), ("g", _fp * (3 * RB_NPARAM)), ("size", ctypes.c_int * (3 * RB_NPARAM)),
                ("c", ctypes.c_int * 4), ("stride", ctypes.c_int * 3), ("n", ctypes.c_int), ("partial", _fp),
                ("rows", ctypes.c_int)]


class ChunkParams(ctypes.Structure):
    _fields_ = [("w", _fp * ST_NPARAM), ("g", _fp * ST_NPARAM), ("size", ctypes.c_int * ST_NPARAM),
                ("ca", ctypes.c_int), ("cs", ctypes.c_int), ("hq", ctypes.c_int), ("c_mid", ctypes.c_int),
                ("delta", ctypes.c_float), ("mul", ctypes.c_float), ("add", ctypes.c_float), ("partial", _fp),
                ("rows", ctypes.c_int)]


class AdamParams(ctypes.Structure):
    _fields_ = [("m", _fp), ("v", _fp), ("step", _fp), ("ticket", _fp), ("lr", _fp), ("beta1", ctypes.c_float),
                ("beta2", ctypes.c_float), ("eps", ctypes.c_float)]


class ChunkSpan(ctypes.Structure):
    _fields_ = [("k0", ctypes.c_int), ("k1", ctypes.c_int), ("s", ctypes.c_int), ("lstates_t", _fp), ("h0", _fp), ("c0", _fp),
                ("hc_bstride", ctypes.c_int), ("dlstates_t", _fp)]


MAX_SPANS = 4
_EP, _CP, _i = ctypes.POINTER(EncoderParams), ctypes.POINTER(ChunkParams), ctypes.c_int
_AP = ctypes.POINTER(AdamParams)
SYMBOLS = (
    ("sur_geometry_supported", [_EP, _EP, _CP]),
    ("sur_encoder_saved_floats", [_EP]),
    ("sur_encoder_forward", [_fp, _EP, _fp, _i, _fp, _fp]),
    ("sur_encoder_backward", [_fp, _EP, _fp, _fp, _i, _fp, _i, _i, _fp]),
    ("sur_encoder_workspace_floats", [_EP, _i]),
    ("sur_encoder_forward_multi", [_fp, _i, ctypes.POINTER(_EP), ctypes.POINTER(_fp), ctypes.POINTER(_i), ctypes.POINTER(_fp),
                                   ctypes.POINTER(_fp), _i]),
    ("sur_encoder_backward_multi", [_fp, _i, ctypes.POINTER(_EP), ctypes.POINTER(_fp), ctypes.POINTER(_fp),
                                    ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_fp),
                                    ctypes.POINTER(_fp)]),
    ("sur_flush_encoder_grads", [_fp, _EP, _AP, _i]),
    ("sur_chunk_saved_floats", [_CP]),
    ("sur_chunk_forward", [_fp, _CP, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _fp]),
    ("sur_chunk_workspace_floats", [_CP, _i, _i]),
    ("sur_chunk_backward", [_fp, _CP, _fp, _fp, _fp, _fp, _i, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _fp, _fp, _fp, _fp,
                            _i, _i, _fp, _fp]),
    ("sur_chunks_backward", [_fp, _CP, _i, ctypes.POINTER(ChunkSpan), _fp, _fp, _fp, _fp, _i, _i, _fp, _i, _i, _fp, _fp]),
    ("sur_flush_chunk_grads", [_fp, _CP, _AP, _i]),
    ("sur_flush_all_grads", [_fp, _EP, _AP, _EP, _AP, _CP, _AP, _i]),
    ("sur_adam_apply", [_fp, _EP, _AP, _EP, _AP, _CP, _AP]),
    ("sur_tbptt_delta_loss", [_fp, _fp, ctypes.c_long, ctypes.c_long, _fp, _i, _i, _i, ctypes.c_float, ctypes.c_float, ctypes.c_float, _fp, _fp, _fp, _fp,
                              _fp, _fp, _fp]),
    ("sur_chunk_integrate", [_fp, _CP, _fp, _fp, _i, _i, _i, _fp]),
    ("sur_tbptt_delta_loss_rows", [_fp, _fp, ctypes.c_long, ctypes.c_long, _fp, _i, _i, _i, ctypes.c_float, ctypes.c_float,
                                   ctypes.c_float, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i]),
    ("sur_tbptt_delta_loss_finalize", [_fp, _i, _i, _i, _fp, _fp, _fp, _fp, _fp]),
    ("sur_fold_rows", [_fp, _EP, _EP, _CP, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    ("sur_tbptt_delta_loss_range", [_fp, _fp, ctypes.c_long, ctypes.c_long, _fp, _i, _i, _i, ctypes.c_float, ctypes.c_float,
                                    ctypes.c_float, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i]),
)
_lib = None


class SurrogateHipError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SurrogateHipError(f"{LIB_PATH} not found: build it (python -c 'import __graft_entry__ as g; "
                                    f"g.build()').  The fused surrogate path has no fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, args in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = ctypes.c_int, args
        lib.sur_last_error.restype = ctypes.c_char_p
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise SurrogateHipError(f"libsurrogate_hip error {rc}: {load().sur_last_error().decode(errors='replace')}")


def _stream():
    # raw handle of torch's current stream; torch.cuda.current_stream() costs ~8 us of Python per call and the eager
    # step makes a dozen
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


# ---------------------------------------------------------------------------------------------
# parameter packs
# ---------------------------------------------------------------------------------------------
class _Pack:
    """Pointers to the weights / gradient tensors of one module in the kernel's order, plus the
    [rows][sum(size)] partial-gradient buffer the backward kernels accumulate into.  ``dirty`` is set
    by a backward launch and cleared by ``flush`` (the reduction of the rows into ``param.grad``).

    Lifetime rules (each one a bug once):
      * gradient tensors are resolved when the flush is QUEUED, not during the forward pass: the caller may run
        ``optimizer.zero_grad(set_to_none=True)`` between forward and backward (pytorch-lightning's closure does);
      * when every ``param.grad`` is undefined the flush writes (g = sum) into a pack-owned flat buffer and hands out
        views of it -- no zero-fill, no allocation per step;
      * a partial buffer that has been superseded by a larger one is kept alive: captured hipGraphs (and launches
        still in flight on side streams) carry its address by value and keep using it consistently;
      * the Adam moments / step counter belong to the parameter set, not to a captured graph: every graph (one per
        batch shape) of the same surrogate is lent the same descriptor."""

    def __init__(self, params, cstruct, flush_fn, rows):
        self.params, self.c, self._flush_fn = params, cstruct, flush_fn
        for i, p in enumerate(params):
            self.c.size[i] = p.numel()
        self.psize = sum(p.numel() for p in params)
        self.partial = None
        self._retired = []
        self.dirty = False
        self.gflat, self._gviews = None, None
        self.overwrite = False    # decided by resolve_grads() for the next flush
        self._adam_state = None   # persistent (m, v, step, ticket)
        self.adam = None          # (AdamParams, tensors kept alive) while the optimizer step runs inside the flush
        self.ensure_rows(rows)
        self.refresh()

    def ensure_rows(self, rows):
        """Make rows [0, rows) available.  The allocation grows geometrically (and superseded buffers are retired, not
        freed); ``c.rows`` -- what the backward launches are checked against and what the flush reduces and re-zeroes --
        is the largest row count ever REQUESTED, so the reduction never walks allocation slack."""
        self.need_rows = max(getattr(self, "need_rows", 0), int(rows))
        if self.partial is None or self.partial.shape[0] < self.need_rows:
            assert not self.dirty, "cannot grow the partial-gradient buffer with unflushed gradients"
            alloc = self.need_rows
            if self.partial is not None:
                self._retired.append(self.partial)   # addresses baked into captured graphs stay valid (and zeroed)
                alloc = max(alloc, (3 * self.partial.shape[0]) // 2)
            alloc = (alloc + 127) // 128 * 128
            self.partial = torch.zeros((alloc, self.psize), device=self.params[0].device, dtype=torch.float32)
        self.c.partial, self.c.rows = self.partial.data_ptr(), self.need_rows

    def refresh(self):
        """Weight addresses (forward pass)."""
        for i, p in enumerate(self.params):
            assert p.is_cuda and p.is_contiguous() and p.dtype == torch.float32
            self.c.w[i] = p.data_ptr()

    def grads_are_own_views(self):
        """True when every param.grad still IS the view of the pack-owned flat buffer handed out by the last flush (the
        kernel's gradient pointers are then current)."""
        views = self._gviews
        return views is not None and all(p.grad is v for p, v in zip(self.params, views))

    def own_flat_grads(self):
        """Point the kernel's gradient addresses at the pack-owned flat buffer (created on first use); param.grad untouched."""
        if self.gflat is None:
            self.gflat = torch.zeros(self.psize, device=self.params[0].device, dtype=torch.float32)
            self._gviews, off = [], 0
            for p in self.params:
                self._gviews.append(self.gflat[off:off + p.numel()].view_as(p))
                off += p.numel()
        for i, view in enumerate(self._gviews):
            self.c.g[i] = view.data_ptr()

    def resolve_grads(self):
        """Gradient addresses, taken when the reduction is about to be launched.  Sets ``self.overwrite``."""
        grads = [p.grad for p in self.params]
        if all(g is None for g in grads):
            if self.gflat is None:
                self.own_flat_grads()
            for i, (p, view) in enumerate(zip(self.params, self._gviews)):
                p.grad = view
                self.c.g[i] = view.data_ptr()
            self.overwrite = True
            return
        for i, (p, g) in enumerate(zip(self.params, grads)):
            if g is None:
                g = p.grad = torch.zeros_like(p)
            if not (g.is_cuda and g.is_contiguous() and g.dtype == torch.float32):
                raise SurrogateHipError("fused backward needs contiguous fp32 CUDA gradient tensors")
            self.c.g[i] = g.data_ptr()
        self.overwrite = False

    def flush(self):
        if self.dirty:
            self.resolve_grads()
            _check(self._flush_fn(_stream(), ctypes.byref(self.c), None if self.adam is None else ctypes.byref(self.adam[0]),
                                  int(self.overwrite and self.adam is None)))
            self.dirty = False

    def adam_descriptor(self, lr_dev, betas=(0.9, 0.999), eps=1e-8):
        """torch.optim.Adam(lr, betas, eps) applied by the flush launch itself.  The moments and the step counter are
        created once per parameter set (zero moments, step 0) and shared by every descriptor handed out."""
        if self._adam_state is None:
            dev = self.params[0].device
            m = torch.zeros(self.psize, device=dev, dtype=torch.float32)
            self._adam_state = (m, torch.zeros_like(m), torch.zeros(1, device=dev, dtype=torch.int32),
                                torch.zeros(1, device=dev, dtype=torch.int32))
        m, v, step, ticket = self._adam_state
        desc = AdamParams(m.data_ptr(), v.data_ptr(), step.data_ptr(), ticket.data_ptr(), lr_dev.data_ptr(), betas[0], betas[1],
                          eps)
        return (desc, m, v, step, ticket, lr_dev)

    def reset_adam(self):
        if self._adam_state is not None:
            for t in self._adam_state:
                t.zero_()

    def disable_adam(self):
        self.adam = None


# one partial row per workgroup of the encoder backward: several workgroups per CU (the block kernels are built for
# 4 waves per SIMD) hide each other's dependent-phase latency
ENCODER_ROWS = 768
# partial rows of one chunk backward launch: the parallel decoder backward runs one workgroup per row
CHUNK_ROWS = 384


def _encoder_pack(convnet, n):
    from pdecontrol.surrogates.models.cnn import ResidualBlock
    blocks = [getattr(convnet, name) for name in getattr(convnet, "layers", ())]
    if len(blocks) != 3 or not all(isinstance(b, ResidualBlock) for b in blocks):
        raise SurrogateHipError("fused encoder expects a ConvNet of three ResidualBlocks (KSAutoRegConvolutionalLSTM family); "
                                "other architectures run on plain PyTorch-ROCm only after ops.enable_fused(False)")
    params, chans, strides = [], [blocks[0].conv3x3_l1.in_channels], []
    for b in blocks:
        if not isinstance(b.activation, nn.SiLU) or b.conv3x3_l1.kernel_size != (3,) or b.conv3x3_l1.bias is not None \
                or b.conv3x3_l1_norm is None or b.conv3x3_l1.padding_mode != "circular":
            raise SurrogateHipError("fused encoder expects SiLU / k=3 / circular / bias-free residual blocks with LayerNorm")
        params += [b.conv3x3_l1.weight, b.conv3x3_l1_norm.weight, b.conv3x3_l1_norm.bias, b.conv3x3_l2.weight,
                   b.conv3x3_l2_norm.weight, b.conv3x3_l2_norm.bias, b.skip.weight, b.skip_norm.weight,
                   b.skip_norm.bias]
        chans.append(b.conv3x3_l1.out_channels)
        strides.append(b.conv3x3_l1.stride[0])
    c = EncoderParams()
    c.c[:] = chans
    c.stride[:] = strides
    c.n = n
    return _Pack(params, c, load().sur_flush_encoder_grads, ENCODER_ROWS)


def _dscale_constants(dscaling):
    """(mul, add) such that dscaling(d) == d * mul + add, for the two forms the controller builds
    (mbrl.py:168-171): identity, or the inverse of a Normalize with scalar statistics."""
    from pdegym.common import transforms as T
    inner = getattr(dscaling, "transform", None)
    if isinstance(dscaling, T.BatchTransform) and isinstance(inner, T.Identity):
        return 1.0, 0.0
    if isinstance(dscaling, T._BatchInverse):
        view = dscaling.transform
        norm = getattr(view, "transf", None)
        if isinstance(norm, T.Normalize) and norm.mean is not None and norm.mean.numel() == 1:
            var, mean = float(norm.var.reshape(-1)[0]), float(norm.mean.reshape(-1)[0])
            mul = float(torch.sqrt(torch.tensor(var, dtype=torch.float32) + norm.epsilon))
            return mul, mean
    raise SurrogateHipError("fused rollout supports dscaling = identity or Normalize(scalar stats).Inverse only")


def scaling_signature(surrogate, undscaling=None):
    """What the scaling constants baked into captured launches depend on.  The controller re-fits the delta statistics
    between training rounds (``update_delta_transform``, pdecontrol/mbrl/mbrl.py:597-602: ``Normalize.reset()`` +
    ``update()`` on the SAME object that the surrogate's ``dscaling`` and the module's ``undscaling`` wrap), so a captured
    graph carries stale (mean, std) afterwards: every graph cache compares this signature (``same_signature``) and
    re-captures when it has changed.  Tensors are held by reference and compared by identity (``update`` assigns new ones)."""
    from pdegym.common import transforms as T
    sig = []
    for tr in (getattr(surrogate, "dscaling", None), undscaling):
        inner = getattr(tr, "transform", None)
        norm = inner if isinstance(inner, T.Normalize) else getattr(inner, "transf", None)
        if isinstance(norm, T.Normalize):
            sig.append((norm, norm.mean, norm.var, norm.count, norm.epsilon))
        else:
            sig.append((type(tr).__name__, type(inner).__name__))
    return sig


def same_signature(a, b):
    if a is None or b is None or len(a) != len(b):
        return False
    for x, y in zip(a, b):
        if len(x) != len(y):
            return False
        for u, v in zip(x, y):
            if isinstance(u, torch.Tensor) or isinstance(v, torch.Tensor) or not isinstance(u, (int, float, str)):
                if u is not v:
                    return False
            elif u != v:
                return False
    return True


def undscale_constants(undscaling):
    """(mean, std) such that undscaling(x) == (x - mean) / std for the two forms the controller builds
    (mbrl.py:168-171): identity, or a Normalize with scalar statistics.  None for anything else."""
    from pdegym.common import transforms as T
    inner = getattr(undscaling, "transform", None)
    if isinstance(undscaling, T.BatchTransform) and isinstance(inner, T.Identity):
        return 0.0, 1.0
    if isinstance(undscaling, T.BatchTransform) and isinstance(inner, T.Normalize) and inner.mean is not None \
            and inner.mean.numel() == 1:
        var = inner.var.reshape(-1)[0].to(torch.float32).cpu()
        return float(inner.mean.reshape(-1)[0]), float(torch.sqrt(var + inner.epsilon))
    return None


def _chunk_pack(surrogate, rows):
    from pdecontrol.surrogates.models.cnn import ConvBlock, DeConvolutionBlock
    from pdecontrol.surrogates.transition import CNNLSTMTransitionModel
    tm = surrogate.transition_model
    if not isinstance(tm, CNNLSTMTransitionModel):
        raise SurrogateHipError("fused chunk expects a CNNLSTMTransitionModel")
    cell, dec = tm.cnnlstmcell, surrogate.state_decoder.model
    blocks = [getattr(dec, name) for name in getattr(dec, "layers", ())]
    ok = (len(blocks) == 4 and isinstance(blocks[0], DeConvolutionBlock) and isinstance(blocks[1], DeConvolutionBlock)
          and isinstance(blocks[2], ConvBlock) and isinstance(blocks[3], ConvBlock)
          and blocks[2].convolution.kernel_size == (7,) and blocks[3].convolution.kernel_size == (5,)
          and isinstance(blocks[3].activation, nn.Identity) and blocks[3].layernorm is None
          and all(isinstance(b.activation, nn.SiLU) for b in blocks[:3]))
    if not ok:
        raise SurrogateHipError("fused chunk expects the KSAutoRegConvolutionalLSTM decoder layout")
    params = []
    for gate in "ifco":
        wx, wh = getattr(cell, f"Wx{gate}"), getattr(cell, f"Wh{gate}")
        params += [wx.weight, wx.bias, wh.weight]
    d0, d1, c2, c3 = blocks
    params += [d0.deconvolution.weight, d0.deconvolution.bias, d0.layernorm.weight, d0.layernorm.bias,
               d1.deconvolution.weight, d1.deconvolution.bias, d1.layernorm.weight, d1.layernorm.bias,
               c2.convolution.weight, c2.convolution.bias, c2.layernorm.weight, c2.layernorm.bias,
               c3.convolution.weight, c3.convolution.bias]
    c = ChunkParams()
    c.ca, c.cs, c.hq = cell.in_channels, cell.out_channels, tm.ssize
    c.c_mid = d1.deconvolution.out_channels
    c.delta = float(surrogate.delta)
    c.mul, c.add = _dscale_constants(surrogate.dscaling)
    return _Pack(params, c, load().sur_flush_chunk_grads, rows)


class FusedPacks:
    def __init__(self, surrogate, n, batch):
        load()
        self.n = n
        self.state_enc = _encoder_pack(surrogate.state_encoder.model, n)
        self.action_enc = _encoder_pack(surrogate.action_encoder.model, n)
        self.chunk = _chunk_pack(surrogate, batch)
        dev = self.chunk.params[0].device
        self.anchor = torch.zeros((), device=dev, requires_grad=True)
        self.key = self._key(surrogate, n)
        self._flush_queued = False
        self.loss_scratch = {}
        self.lr_dev = torch.zeros(1, device=dev, dtype=torch.float32)   # Adam's learning rate, read by the flush launch
        self._lr_host = None

    @staticmethod
    def _key(surrogate, n):
        p = next(surrogate.parameters())
        return (p.data_ptr(), n)

    @property
    def packs(self):
        return (self.state_enc, self.action_enc, self.chunk)

    def refresh(self, surrogate, batch):
        self.chunk.ensure_rows(batch)
        for pack in self.packs:
            pack.refresh()
        self.chunk.c.mul, self.chunk.c.add = _dscale_constants(surrogate.dscaling)

    def refresh_partials(self):
        for pack in self.packs:
            pack.c.partial, pack.c.rows = pack.partial.data_ptr(), pack.need_rows

    def flush(self):
        """Reduce every pending partial-gradient row into param.grad (3 tiny launches at most)."""
        self._flush_queued = False
        dirty = [pack for pack in self.packs if pack.dirty]
        if not dirty:
            return
        for pack in dirty:
            pack.resolve_grads()
        if len(dirty) == 3:   # the usual case: one launch reduces (and, with Adam enabled, applies) all three packs
            ad = lambda pack: None if pack.adam is None else ctypes.byref(pack.adam[0])
            mask = sum((1 << j) for j, pack in enumerate(self.packs) if pack.overwrite and pack.adam is None)
            _check(load().sur_flush_all_grads(_stream(), ctypes.byref(self.state_enc.c), ad(self.state_enc),
                                              ctypes.byref(self.action_enc.c), ad(self.action_enc),
                                              ctypes.byref(self.chunk.c), ad(self.chunk), mask))
            for pack in dirty:
                pack.dirty = False
            return
        for pack in dirty:
            pack.flush()

    def flush_into_flat(self, accumulate):
        """Reduce the partial rows into the packs' OWN flat gradient buffers (``gflat``; overwrite, or add when
        ``accumulate``), whatever ``param.grad`` currently is: the captured backward of the split-graph step, whose caller
        re-attaches the views to ``param.grad`` after each replay (``hand_out_flat_grads``)."""
        self._flush_queued = False
        for pack in self.packs:
            pack.own_flat_grads()
            pack.dirty = False
        _check(load().sur_flush_all_grads(_stream(), ctypes.byref(self.state_enc.c), None, ctypes.byref(self.action_enc.c), None,
                                          ctypes.byref(self.chunk.c), None, 0 if accumulate else 7))

    # -- optimizer inside the flush launches ---------------------------------------------------------------------
    def set_lr(self, lr):
        """Learning rate of the in-kernel Adam: a device scalar, so a scheduler can move it between graph replays."""
        lr = float(lr)
        if lr != self._lr_host:
            self.lr_dev.fill_(lr)
            self._lr_host = lr

    def adam_descriptors(self, lr, betas=(0.9, 0.999), eps=1e-8):
        """One descriptor per pack over the surrogate's ONE Adam state (created on first use)."""
        self.set_lr(lr)
        return [pack.adam_descriptor(self.lr_dev, betas, eps) for pack in self.packs]

    def adam_step_count(self):
        st = self.chunk._adam_state
        return 0 if st is None else int(st[2].item())

    def reset_adam(self):
        for pack in self.packs:
            pack.reset_adam()

    def lend_adam(self, descriptors):
        for pack, d in zip(self.packs, descriptors):
            pack.adam = d

    def disable_adam(self):
        for pack in self.packs:
            pack.disable_adam()

    def schedule_flush(self):
        """Called from inside a backward: run ``flush`` when the current backward pass finishes."""
        if not self._flush_queued:
            self._flush_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)


class PackAdam(torch.optim.Optimizer):
    """``torch.optim.Adam(lr, betas, eps)`` (no weight decay / amsgrad) for the parameters of ONE fused surrogate: the
    whole update is a single launch (``sur_adam_apply``) over the packs' flat moment buffers -- the optimizer
    ``PDETrainingModule.configure_optimizers`` hands to ``pl.Trainer.fit`` on a GPU.  torch's own Adam spends ~0.4 ms of
    Python per step on 62 small parameters, as much as the whole fused forward + backward takes on the device.

    * the Adam state (moments, step counter) is the surrogate's one state, shared with the captured-graph step;
    * ``param_groups[0]["lr"]`` is honoured every step (schedulers work);
    * packs whose gradients are all undefined are skipped, like torch skips parameters without ``.grad``;
    * with an initialised ``torch.distributed`` group of more than one rank the pack gradients are averaged over the
      ranks first (flat buffers: three small all-reduces) -- the fused backward writes ``param.grad`` outside
      autograd, so hook-based DDP wrappers never see these gradients."""

    def __init__(self, surrogate, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = [p for p in surrogate.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._surrogate = surrogate

    def _packs(self):
        packs = getattr(self._surrogate, "_fused_packs", None)
        if packs is None:
            raise SurrogateHipError("PackAdam.step() before any fused forward / backward of its surrogate")
        return packs

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        packs = self._packs()
        group = self.param_groups[0]
        pending = self.__dict__.pop("_pending_state", None)
        if pending is not None:
            self._apply_pack_state(packs, pending)
        key = (id(packs), tuple(group["betas"]), group["eps"])
        if self.__dict__.get("_desc_key") != key:       # descriptors are built once; only the device lr changes per step
            self._descs, self._desc_key = packs.adam_descriptors(group["lr"], group["betas"], group["eps"]), key
        packs.set_lr(group["lr"])
        live = []
        for pack, desc in zip(packs.packs, self._descs):
            if pack.grads_are_own_views():               # the usual case: what the flush of this step just handed out
                live.append(desc)
            elif all(p.grad is None for p in pack.params):
                live.append(None)
            else:
                pack.resolve_grads()
                live.append(desc)
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            world = torch.distributed.get_world_size()
            for pack, desc in zip(packs.packs, live):
                if desc is None:
                    continue
                if pack.gflat is None or pack.params[0].grad.data_ptr() != pack.gflat.data_ptr():
                    raise SurrogateHipError("PackAdam data-parallel averaging needs the pack-owned flat gradients "
                                            "(zero_grad(set_to_none=True) before each backward)")
                torch.distributed.all_reduce(pack.gflat)
                pack.gflat.div_(world)
        ref = lambda d: None if d is None else ctypes.byref(d[0])
        se, ae, ch = packs.packs
        _check(load().sur_adam_apply(_stream(), ctypes.byref(se.c), ref(live[0]), ctypes.byref(ae.c), ref(live[1]),
                                     ctypes.byref(ch.c), ref(live[2])))
        return loss

    def state_dict(self):
        out = {"param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups], "packs": []}
        packs = getattr(self._surrogate, "_fused_packs", None)
        if packs is not None:
            for pack in packs.packs:
                st = pack._adam_state
                out["packs"].append(None if st is None else {"exp_avg": st[0].clone(), "exp_avg_sq": st[1].clone(),
                                                              "step": int(st[2].item())})
        return out

    def load_state_dict(self, state):
        for g, saved in zip(self.param_groups, state.get("param_groups", [])):
            g.update(saved)
        if not state.get("packs"):
            return
        packs = getattr(self._surrogate, "_fused_packs", None)
        if packs is None:
            # a checkpoint is restored before the first forward pass, i.e. before the packs exist: keep the moments and
            # hand them over when the first step() finds the packs (dropping them would restart Adam from zero, silently)
            self._pending_state = [None if s is None else {k: (v.clone() if torch.is_tensor(v) else v) for k, v in s.items()}
                                   for s in state["packs"]]
            return
        self._apply_pack_state(packs, state["packs"])

    def _apply_pack_state(self, packs, saved_packs):
        if len(saved_packs) != len(packs.packs):
            raise SurrogateHipError(f"PackAdam state has {len(saved_packs)} packs, the surrogate {len(packs.packs)}")
        group = self.param_groups[0]
        packs.adam_descriptors(group["lr"], group["betas"], group["eps"])
        for pack, saved in zip(packs.packs, saved_packs):
            if saved is None:
                continue
            if saved["exp_avg"].numel() != pack._adam_state[0].numel():
                raise SurrogateHipError("PackAdam state does not fit this surrogate (different parameter count)")
            pack._adam_state[0].copy_(saved["exp_avg"])
            pack._adam_state[1].copy_(saved["exp_avg_sq"])
            pack._adam_state[2].fill_(int(saved["step"]))


def fused_supported(surrogate):
    """True when ``surrogate`` has the KSAutoRegConvolutionalLSTM layout the fused kernels implement."""
    try:
        from pdecontrol.surrogates.models.cnn import ResidualBlock
        from pdecontrol.surrogates.surrogate import AutoRegPDESurrogate
        from pdecontrol.surrogates.transition import CNNLSTMTransitionModel
        if not isinstance(surrogate, AutoRegPDESurrogate) or not isinstance(surrogate.transition_model, CNNLSTMTransitionModel):
            return False
        for enc in (surrogate.state_encoder.model, surrogate.action_encoder.model):
            blocks = [getattr(enc, name) for name in getattr(enc, "layers", ())]
            if len(blocks) != 3 or not all(isinstance(b, ResidualBlock) for b in blocks):
                return False
        return len(getattr(surrogate.state_decoder.model, "layers", ())) == 4
    except Exception:
        return False


_GEOMETRY = {}


def geometry_unsupported(surrogate, n):
    """None when the fused kernels implement grid width ``n`` for this (architecturally supported) surrogate, else the
    library's reason.  The rule lives in the library (``sur_geometry_supported``: every LayerNorm row must be 16, 32 or a
    multiple of 64 up to 256 values wide -- N in {64, 128, 256} with the reference's strides); every launch checks it too."""
    strides = tuple(tuple(int(getattr(enc.model, name).conv3x3_l1.stride[0]) for name in enc.model.layers)
                    for enc in (surrogate.state_encoder, surrogate.action_encoder))
    key = (int(n), strides, int(surrogate.transition_model.ssize))
    if key not in _GEOMETRY:
        lib = load()
        encs = []
        for st in strides:
            c = EncoderParams()
            c.n = int(n)
            c.stride[:] = st
            encs.append(c)
        ch = ChunkParams()
        ch.hq = int(surrogate.transition_model.ssize)
        rc = lib.sur_geometry_supported(ctypes.byref(encs[0]), ctypes.byref(encs[1]), ctypes.byref(ch))
        _GEOMETRY[key] = None if rc == 0 else lib.sur_last_error().decode()
    return _GEOMETRY[key]


def packs_for(surrogate, n, batch):
    packs = getattr(surrogate, "_fused_packs", None)
    if packs is None or packs.key != FusedPacks._key(surrogate, n):
        packs = FusedPacks(surrogate, n, batch)
        object.__setattr__(surrogate, "_fused_packs", packs)
    else:
        packs.refresh(surrogate, batch)
    return packs


# ---------------------------------------------------------------------------------------------
# autograd Functions
# ---------------------------------------------------------------------------------------------
class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, pack, owner):
        x = x.contiguous()
        m, h = x.shape[0], pack.c.n
        for s in pack.c.stride:
            h //= s
        z = torch.empty((m, pack.c.c[3], h), device=x.device, dtype=torch.float32)
        saved = _encoder_saved_buffer(pack, m, x.device) if any(ctx.needs_input_grad) else None
        _check(load().sur_encoder_forward(_stream(), ctypes.byref(pack.c), _p(x), m, _p(z), _p(saved)))
        ctx.save_for_backward(x)
        ctx.pack, ctx.owner, ctx.need_dx, ctx.fwd_saved = pack, owner, x.requires_grad, saved
        return z

    @staticmethod
    def backward(ctx, dz):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x) if ctx.need_dx else None
        if dx is None and ctx.fwd_saved is not None:   # raw-data input: block-per-launch backward
            _encoder_backward_multi(load(), [(ctx.pack, x, dz.contiguous(), x.shape[0], 0,
                                              min(ENCODER_ROWS, ctx.pack.c.rows), ctx.fwd_saved)])
        else:
            _check(load().sur_encoder_backward(_stream(), ctypes.byref(ctx.pack.c), _p(x), _p(dz.contiguous()), x.shape[0],
                                               _p(dx), 0, min(ENCODER_ROWS, ctx.pack.c.rows), _p(ctx.fwd_saved)))
        ctx.pack.dirty = True
        ctx.owner.schedule_flush()
        return dx, None, None, None


SAVE_ACTIVATIONS = True   # False: the ENCODER backward recomputes its forward (bit-identical, slower; test hook)


def _encoder_saved_buffer(pack, m, device):
    """[M, F] buffer for the encoder's forward intermediates (None: no saved-activation path for this geometry)."""
    f = load().sur_encoder_saved_floats(ctypes.byref(pack.c)) if SAVE_ACTIVATIONS else 0
    return torch.empty((m, f), device=device, dtype=torch.float32) if f > 0 else None


def _chunk_workspace(pack, k, b, device):
    """Scratch of the chunk backward's split path (per-step decoder gradients wrt h and wrt the predicted deltas)."""
    f = load().sur_chunk_workspace_floats(ctypes.byref(pack.c), k, b)
    return torch.empty(f, device=device, dtype=torch.float32)


def _saved_buffer(pack, k, b, device):
    """[K, B, F] buffer for the forward intermediates of a chunk: the backward kernels (parallel decoder backward +
    cell chain) read them back instead of recomputing the steps."""
    f = load().sur_chunk_saved_floats(ctypes.byref(pack.c))
    if f <= 0:
        raise SurrogateHipError("the fused chunk kernels need a latent width N/4 that is a multiple of 16 (N = 64, 128, 256)")
    return torch.empty((k, b, f), device=device, dtype=torch.float32)


class _ChunkFn(torch.autograd.Function):
    """K rollout steps in one launch; inputs / outputs are time-major (see surrogate_hip.h)."""

    @staticmethod
    def forward(ctx, xlat_t, lstates_t, states_t, h0, c0, anchor, pack, owner):
        xlat_t, lstates_t, states_t = xlat_t.contiguous(), lstates_t.contiguous(), states_t.contiguous()
        h0, c0 = h0.contiguous(), c0.contiguous()
        k, b = xlat_t.shape[:2]
        s = lstates_t.shape[0]
        n = states_t.shape[-1]
        h_all = torch.empty((k, b, pack.c.cs, pack.c.hq), device=xlat_t.device, dtype=torch.float32)
        c_all = torch.empty_like(h_all)
        d_all = torch.empty((k, b, 1, n), device=xlat_t.device, dtype=torch.float32)
        out_all = torch.empty_like(d_all)
        saved = _saved_buffer(pack, k, b, xlat_t.device) if any(ctx.needs_input_grad) else None
        _check(load().sur_chunk_forward(_stream(), ctypes.byref(pack.c), _p(xlat_t), _p(lstates_t), _p(states_t), _p(h0),
                                        _p(c0), pack.c.cs * pack.c.hq, k, s, b, _p(h_all), _p(c_all), _p(d_all), _p(out_all),
                                        _p(saved)))
        ctx.save_for_backward(xlat_t, lstates_t, h0, c0, h_all, c_all)
        ctx.fwd_saved = saved
        ctx.pack, ctx.owner = pack, owner
        ctx.needs = (xlat_t.requires_grad, lstates_t.requires_grad, h0.requires_grad, c0.requires_grad)
        ctx.set_materialize_grads(False)
        return h_all, c_all, d_all, out_all

    @staticmethod
    def backward(ctx, dh_all, dc_all, dd_all, dout_all):
        xlat_t, lstates_t, h0, c0, h_all, c_all = ctx.saved_tensors
        nx, nl, nh, nc = ctx.needs
        cont = lambda t: None if t is None else t.contiguous()
        dh_all, dc_all, dd_all, dout_all = cont(dh_all), cont(dc_all), cont(dd_all), cont(dout_all)
        dxlat = torch.empty_like(xlat_t) if nx else None
        dlst = torch.zeros_like(lstates_t) if nl else None
        dh0 = torch.empty_like(h0) if nh else None
        dc0 = torch.empty_like(c0) if nc else None
        k, b = xlat_t.shape[:2]
        rows = max(b, CHUNK_ROWS)
        ctx.pack.ensure_rows(rows)
        ctx.owner.refresh_partials()
        work = _chunk_workspace(ctx.pack, k, b, xlat_t.device) if ctx.fwd_saved is not None else None
        _check(load().sur_chunk_backward(_stream(), ctypes.byref(ctx.pack.c), _p(xlat_t), _p(lstates_t), _p(h0), _p(c0),
                                         ctx.pack.c.cs * ctx.pack.c.hq, _p(h_all), _p(c_all), _p(dd_all), _p(dout_all), _p(dh_all), _p(dc_all), k,
                                         lstates_t.shape[0], b, _p(dxlat), _p(dlst), _p(dh0), _p(dc0), 0, rows,
                                         _p(ctx.fwd_saved), _p(work)))
        ctx.pack.dirty = True
        ctx.owner.schedule_flush()
        return dxlat, dlst, None, dh0, dc0, None, None, None


class _DeltaLossFn(torch.autograd.Function):
    """Delta-mode TBPTT loss, its time-resolved mean, the logged statistics and d loss / d d_all in one launch
    (sur_tbptt_delta_loss, include/surrogate_hip.h)."""

    @staticmethod
    def forward(ctx, d_all, states, delta, mean, stdv, scratch):
        t, b, _, n = d_all.shape
        dev = d_all.device
        d_all = d_all.contiguous()
        if states.stride(3) != 1:          # rows of N must be dense; batch / time strides are free (C = 1)
            states = states.contiguous()
        new = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
        deltas, hstep, loss, stats = new(b, t - 1, 1, n), new(t - 1), new(), new(4)
        dd = torch.empty_like(d_all) if ctx.needs_input_grad[0] else None
        ctx.set_materialize_grads(False)
        partial, ticket = scratch
        _check(load().sur_tbptt_delta_loss(_stream(), _p(states), states.stride(0), states.stride(1), _p(d_all), b, t, n, delta,
                                           mean, stdv, _p(deltas), _p(dd),
                                           _p(hstep), _p(loss), _p(stats), _p(partial), _p(ticket)))
        ctx.dd = dd
        ctx.mark_non_differentiable(deltas, hstep, stats)
        return loss, hstep, stats, deltas

    @staticmethod
    def backward(ctx, g_loss, *_):
        dd = ctx.dd                      # kept: backward may run again through a retained graph
        if dd is None or g_loss is None:
            return None, None, None, None, None, None
        if g_loss.data_ptr() == _UNIT_GRADS.get(g_loss.device, (None, 0))[1]:
            return dd, None, None, None, None, None      # d loss / d loss = 1 handed in by unit_grad(): no multiply
        return dd * g_loss, None, None, None, None, None


_UNIT_GRADS = {}


def unit_grad(device):
    """A cached scalar 1.0 to start a backward pass from (``loss.backward(gradient=unit_grad(dev))``): saves the
    ones_like fill, and the fused loss recognises it and skips the multiplication by it."""
    device = torch.device(device)
    if device not in _UNIT_GRADS:
        t = torch.ones((), device=device, dtype=torch.float32)
        _UNIT_GRADS[device] = (t, t.data_ptr())
    return _UNIT_GRADS[device][0]


def fused_delta_loss(surrogate, d_all, states, delta, mean, stdv):
    """d_all: time-major predicted deltas [T,B,1,N] (autograd output of the fused TBPTT forward); states [B,T,1,N].
    Returns (loss, hsteploss [T-1], stats [4] = mean/std of predicted then true deltas, true deltas [B,T-1,1,N])."""
    owner = getattr(surrogate, "_fused_packs", None)
    if owner is not None:
        cache = owner.loss_scratch
    else:                       # a surrogate without packs (the FNO path): the scratch hangs on the surrogate itself
        cache = surrogate.__dict__.setdefault("_loss_scratch", {})
    t = d_all.shape[0]
    scratch = cache.get((t, d_all.device))
    if scratch is None:
        scratch = (torch.empty(40 * t, device=d_all.device, dtype=torch.float64),
                   torch.zeros(1, device=d_all.device, dtype=torch.int32))
        cache[(t, d_all.device)] = scratch
    return _DeltaLossFn.apply(d_all, states, float(delta), float(mean), float(stdv), scratch)


def _encoder_forward_multi(lib, jobs):
    """jobs: one or two (pack, x_ptr, m, z_ptr, saved_ptr) tuples (raw device addresses) -> block-per-launch forward."""
    n = len(jobs)
    arr = lambda ctype, vals: (ctype * n)(*vals)
    _check(lib.sur_encoder_forward_multi(
        _stream(), n, arr(_EP, [ctypes.pointer(j[0].c) for j in jobs]), arr(_fp, [ctypes.c_void_p(j[1]) for j in jobs]),
        arr(_i, [j[2] for j in jobs]), arr(_fp, [ctypes.c_void_p(j[3]) for j in jobs]),
        arr(_fp, [ctypes.c_void_p(j[4]) for j in jobs]), ENCODER_ROWS))


def _encoder_backward_multi(lib, jobs):
    """jobs: up to three (pack, x, dz, m, row_base, row_count, saved) tuples -> one sur_encoder_backward_multi launch."""
    n = len(jobs)
    arr = lambda ctype, vals: (ctype * n)(*vals)
    ptr = lambda t: ctypes.c_void_p(None if t is None else t.data_ptr())
    # scratch for the gradients between residual blocks (block-per-launch backward, needs the saved activations)
    works = [None if j[6] is None else torch.empty(lib.sur_encoder_workspace_floats(ctypes.byref(j[0].c), j[3]),
                                                   device=j[1].device, dtype=torch.float32) for j in jobs]
    _check(lib.sur_encoder_backward_multi(
        _stream(), n, arr(_EP, [ctypes.pointer(j[0].c) for j in jobs]), arr(_fp, [ptr(j[1]) for j in jobs]),
        arr(_fp, [ptr(j[2]) for j in jobs]), arr(_i, [j[3] for j in jobs]), arr(_i, [j[4] for j in jobs]),
        arr(_i, [j[5] for j in jobs]), arr(_fp, [ptr(j[6]) for j in jobs]), arr(_fp, [ptr(w) for w in works])))
    for w in works:
        if w is not None:
            w.record_stream(torch.cuda.current_stream(w.device))


def encode(x, pack, owner):
    """[M, C0, N] -> [M, C3, N/4] through the fused 3-block residual encoder."""
    return _EncoderFn.apply(x, owner.anchor, pack, owner)


def rollout_chunk(xlat_t, lstates_t, states_t, h0, c0, owner):
    return _ChunkFn.apply(xlat_t, lstates_t, states_t, h0, c0, owner.anchor, owner.chunk, owner)


def fused_rollout(surrogate, states, actions, times, targets, hidden):
    """GPU rollout of AutoRegPDESurrogate (same outputs as surrogate.py:79-133; ``inlatents`` is not
    produced): two encoder launches + ONE launch for all time steps.  states [B,S,1,N], actions [B,A,1,N]."""
    from pdecontrol.mbrl.types import ModelRollout
    from pdecontrol.surrogates.surrogate import action_and_target_indices, take_steps
    b, s_given, _, n = states.shape
    owner = packs_for(surrogate, n, b)
    cs, hq, ca = owner.chunk.c.cs, owner.chunk.c.hq, owner.chunk.c.ca
    aidx, tidx = action_and_target_indices(times, targets, surrogate.delta)
    # time-major layout: every per-step slice is a contiguous [B, ...] block
    actions_t = take_steps(actions, aidx.tolist()).transpose(0, 1).contiguous()         # [K, B, 1, N]
    n_steps = actions_t.shape[0]
    s_used = min(s_given, n_steps)
    states_t = states[:, :s_used].transpose(0, 1).contiguous()                          # [S, B, 1, N]
    lstates_t = encode(states_t.reshape(s_used * b, 1, n), owner.state_enc, owner).reshape(s_used, b, cs, hq)
    lactions_t = encode(actions_t.reshape(n_steps * b, 1, n), owner.action_enc, owner).reshape(n_steps, b, ca, hq)
    if hidden is None:
        tm = surrogate.transition_model
        hidden = (tm.H0.unsqueeze(0).expand(b, -1, -1), tm.C0.unsqueeze(0).expand(b, -1, -1))
    h_all, c_all, d_all, out_all = rollout_chunk(lactions_t, lstates_t, states_t, hidden[0], hidden[1], owner)
    pick = tidx.tolist()
    by_batch = lambda t: take_steps(t.transpose(0, 1), pick)
    return ModelRollout(inlatents=None, outlatents=by_batch(h_all), deltas=by_batch(d_all), outputs=by_batch(out_all),
                        hidden=(h_all[-1], c_all[-1]))


# ---------------------------------------------------------------------------------------------
# whole TBPTT forward/backward as ONE autograd node with hand-scheduled streams
# ---------------------------------------------------------------------------------------------
_STREAM_POOLS = {}


def pooled_streams(device, n, kind="side"):
    """The first ``n`` streams of a per-device, per-purpose pool that only ever grows.  torch hands out streams
    round-robin from 32 per device, so code that creates a fresh ``torch.cuda.Stream`` per object eventually gets one
    that aliases a stream it forks from or captures on; a handful of long-lived streams cannot.  Kinds in use:
    "side" (forks inside one TBPTT step), "capture" (warm-up + hipGraph capture), "member" (ensemble siblings)."""
    device = torch.device(device)
    pool = _STREAM_POOLS.setdefault((device, kind), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def _side_streams(owner, device, n):
    """Side streams of one TBPTT step.  Steps of different surrogates never fork concurrently outside an ensemble
    capture, and there the members keep to one stream each (``inner_forks(False)``), so the pool is shared."""
    return pooled_streams(device, n, "side")


_INNER_FORKS = True


class inner_forks:
    """``with inner_forks(False): ...`` -- the fused TBPTT step stays on the current stream instead of forking
    side streams.  Needed when the step itself runs on a forked stream of a hipGraph capture: a forked stream
    that forks again crashes hipStreamEndCapture on ROCm 7 (tools/dbg_capture.py)."""

    def __init__(self, enabled):
        self.enabled = bool(enabled)

    def __enter__(self):
        global _INNER_FORKS
        self.prev, _INNER_FORKS = _INNER_FORKS, self.enabled
        return self

    def __exit__(self, *exc):
        global _INNER_FORKS
        _INNER_FORKS = self.prev


class _Fork:
    """``with _Fork(stream): ...`` -- run the block on ``stream`` after everything queued so far on the
    current stream; ``join()`` makes the current stream wait for it.  Works eagerly and under hipGraph
    capture (the side work becomes a parallel branch of the graph).  With ``inner_forks(False)`` the block
    simply runs on the current stream."""

    def __init__(self, stream, worthwhile=True, after=None):
        """``after``: an event recorded earlier on the current stream = the fork point (default: now).  Recording the
        fork point first and entering the block LATER leaves the dependencies unchanged but lets the current stream's own
        next launches be issued -- under capture: created as graph nodes -- before the side work.  ROCm's graph executor
        keeps a node's FIRST-created successor on the predecessor's hardware queue and moves later-created successors to
        another queue, and an edge that changes queues costs ~10 us: the critical path must be captured first."""
        self.main = torch.cuda.current_stream(stream.device)
        self.after = after
        # ``worthwhile`` = False: the caller knows the step is host-bound when run eagerly (small grids: the GPU waits
        # for Python, a fork only adds events and stream switches); under capture the fork always pays (a graph branch)
        self.forked = _INNER_FORKS and (worthwhile or torch.cuda.is_current_stream_capturing())
        self.stream = stream if self.forked else self.main

    def __enter__(self):
        if self.forked:
            if self.after is not None:
                self.stream.wait_event(self.after)
            else:
                self.stream.wait_stream(self.main)
        self.ctx = torch.cuda.stream(self.stream)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        self.ctx.__exit__(*exc)

    def join(self):
        if self.forked:
            self.main.wait_stream(self.stream)


def _tbptt_forward(states, actions, owner, surrogate, tau, tbtt, after_chunk=None, integrate_last=True):
    """The forward launches of a TBPTT pass (see ``_TBPTTFn``).  ``after_chunk(c, st)`` -- if given -- is called right
    after chunk ``c``'s rollout has been queued, with the pass's state so far (``st``: the namespace this function
    returns): the pipelined training pass hangs that chunk's loss + backward branch there.  ``integrate_last`` = False
    leaves the LAST chunk's predictions un-integrated (``st.out_all`` rows of that chunk unwritten): nothing is rolled out
    from them, the caller queues ``sur_chunk_integrate`` off the critical path."""
    b, t_total, _, n = actions.shape
    cs, hq, ca = owner.chunk.c.cs, owner.chunk.c.hq, owner.chunk.c.ca
    dev = actions.device
    lib = load()
    bounds = [(k0, min(k0 + tbtt, t_total)) for k0 in range(0, t_total, tbtt)]
    nchunks = len(bounds)
    (side,) = _side_streams(owner, dev, 1)

    actions_t = actions.transpose(0, 1).contiguous()                    # [T, B, 1, N]
    states_t0 = states[:, :tau].transpose(0, 1).contiguous()            # [tau, B, 1, N]
    lactions_t = torch.empty((t_total, b, ca, hq), device=dev, dtype=torch.float32)
    # Action latents chunk by chunk on the side stream: chunk 0's first (the cell chain waits for them, beside the
    # chunk-0 state encoding), the later chunks' while chunk 0's cell chain occupies only B of the 256 CUs.
    asaved = _encoder_saved_buffer(owner.action_enc, t_total * b, dev)
    af = 0 if asaved is None else asaved.shape[1]
    nin, nlat = actions_t.shape[2] * n, ca * hq
    lstates = [torch.empty((tau, b, cs, hq), device=dev, dtype=torch.float32)]
    ssaved = [_encoder_saved_buffer(owner.state_enc, tau * b, dev)]

    def action_job(k0, k1):
        lo = k0 * b
        return (owner.action_enc, actions_t.data_ptr() + 4 * lo * nin, (k1 - k0) * b, lactions_t.data_ptr() + 4 * lo * nlat,
                None if asaved is None else asaved.data_ptr() + 4 * lo * af)

    split = asaved is not None and ssaved[0] is not None   # block-per-launch encoders need the saved records
    if split:   # chunk-0 state encoding and chunk-0 action latents: one block-per-launch forward, both jobs per launch
        _encoder_forward_multi(lib, [(owner.state_enc, states_t0.data_ptr(), tau * b, lstates[0].data_ptr(),
                                      ssaved[0].data_ptr()), action_job(*bounds[0])])
    # the other action latents on the side stream: with `split`, forked AFTER the launches above (which fill the
    # device anyway) so that they run beside chunk 0's cell chain, which occupies only B of the 256 CUs
    fork_point = None
    if split and nchunks > 1 and _INNER_FORKS and (n >= 128 or torch.cuda.is_current_stream_capturing()):
        fork_point = torch.cuda.Event()
        fork_point.record(torch.cuda.current_stream(dev))
    fork = _Fork(side, worthwhile=n >= 128, after=fork_point)
    lat_ready = {}

    def encode_later_actions():
        with fork:
            for c, (k0, k1) in enumerate(bounds):
                if split and c == 0:
                    continue
                job = action_job(k0, k1)
                if split:
                    _encoder_forward_multi(lib, [job])
                else:
                    _check(lib.sur_encoder_forward(_stream(), ctypes.byref(owner.action_enc.c), ctypes.c_void_p(job[1]), job[2],
                                                   ctypes.c_void_p(job[3]), None if job[4] is None else ctypes.c_void_p(job[4])))
                if fork.forked:
                    ev = torch.cuda.Event()
                    ev.record(fork.stream)
                    lat_ready[c] = ev
            if fork.forked:
                for t in (asaved, lactions_t, actions_t):
                    if t is not None:
                        t.record_stream(torch.cuda.current_stream(dev))

    if fork_point is None:
        encode_later_actions()     # chunk 0's own action latents are among them (or nothing forks): before the chunk loop
    if not split:
        _check(lib.sur_encoder_forward(_stream(), ctypes.byref(owner.state_enc.c), _p(states_t0), tau * b, _p(lstates[0]),
                                       _p(ssaved[0])))
    main = torch.cuda.current_stream(dev)

    tm = surrogate.transition_model
    h0, c0 = tm.H0.detach().contiguous(), tm.C0.detach().contiguous()   # one [cs, hq] state shared by the batch
    s_lat = cs * hq
    seeds, h0s, c0s, h_alls, c_alls, saveds = [states_t0], [h0], [c0], [], [], []
    d_all = torch.empty((t_total, b, 1, n), device=dev, dtype=torch.float32)
    out_all = torch.empty_like(d_all)
    # one time-major tensor per quantity for ALL chunks: the backward pass then runs every chunk in the same launches
    h_all_u = torch.empty((t_total, b, cs, hq), device=dev, dtype=torch.float32)
    c_all_u = torch.empty_like(h_all_u)
    saved_u = _saved_buffer(owner.chunk, t_total, b, dev)
    st = types.SimpleNamespace(b=b, t_total=t_total, n=n, nchunks=nchunks, bounds=bounds, s_lat=s_lat, actions_t=actions_t,
                               lactions_t=lactions_t, seeds=seeds, lstates=lstates, h0s=h0s, c0s=c0s, h_alls=h_alls,
                               c_alls=c_alls, saveds=saveds, asaved=asaved, ssaved=ssaved, d_all=d_all, out_all=out_all,
                               h_all_u=h_all_u, c_all_u=c_all_u, saved_u=saved_u, split=split)
    for c, (k0, k1) in enumerate(bounds):
        if c > 0:   # later chunks restart from the previous chunk's last prediction (gradients cut)
            seeds.append(out_all[k0 - 1:k0])
            lst = torch.empty((1, b, cs, hq), device=dev, dtype=torch.float32)
            ssaved.append(_encoder_saved_buffer(owner.state_enc, b, dev))
            _check(lib.sur_encoder_forward(_stream(), ctypes.byref(owner.state_enc.c), _p(seeds[c]), b, _p(lst),
                                           _p(ssaved[c])))
            lstates.append(lst)
            h0s.append(h_alls[-1][-1])
            c0s.append(c_alls[-1][-1])
        k = k1 - k0
        h_all, c_all, saved = h_all_u[k0:k1], c_all_u[k0:k1], saved_u[k0:k1]
        s_used = min(seeds[c].shape[0], k)
        if c in lat_ready:
            main.wait_event(lat_ready[c])    # this chunk's action latents (encoded on the side stream)
        _check(lib.sur_chunk_forward(_stream(), ctypes.byref(owner.chunk.c), _p(lactions_t[k0:k1]), _p(lstates[c]),
                                     _p(seeds[c]), _p(h0s[c]), _p(c0s[c]), 0 if c == 0 else s_lat, k, s_used, b,
                                     _p(h_all), _p(c_all), _p(d_all[k0:k1]),
                                     _p(out_all[k0:k1]) if (integrate_last or c < nchunks - 1) else None, _p(saved)))
        h_alls.append(h_all)
        c_alls.append(c_all)
        saveds.append(saved)
        if c == 0 and fork_point is not None:
            encode_later_actions()  # forked at `fork_point`, issued after the critical path's own launches (see _Fork)
        if after_chunk is not None:
            after_chunk(c, st)
    fork.join()
    return st


class _TBPTTFn(torch.autograd.Function):
    """All chunks of a truncated-BPTT forward pass (training.py:71-98) and their backward.

    Forward: the state encoder of chunk 0 and the action encoder of ALL T steps run concurrently; then
    chunk kernel, (1-state) encoder, chunk kernel, ...  Backward: TBPTT cuts the graph between chunks, so
    the chunks' backward kernels are independent -- they run concurrently on side streams, each followed
    by its state-encoder backward; the single action-encoder backward joins them.  Concurrent launches
    get disjoint partial-gradient row ranges."""

    @staticmethod
    def forward(ctx, states, actions, anchor, owner, surrogate, tau, tbtt):
        st = _tbptt_forward(states, actions, owner, surrogate, tau, tbtt)
        ctx.owner, ctx.bounds, ctx.dims = owner, st.bounds, (st.b, st.t_total, st.n, st.nchunks)
        ctx.saved = (st.actions_t, st.lactions_t, st.seeds, st.lstates, st.h0s, st.c0s, st.h_alls, st.c_alls, st.saveds, st.asaved,
                     st.ssaved)
        ctx.unified = (st.h_all_u, st.c_all_u, st.saved_u)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(st.out_all)
        return st.d_all, st.out_all, st.h_alls[-1][-1], st.c_alls[-1][-1]

    @staticmethod
    def backward(ctx, dd_all, _dout, _dh, _dc):
        if dd_all is None:
            return (None,) * 7
        _tbptt_backward(ctx.owner, ctx.bounds, ctx.dims, ctx.saved, ctx.unified, dd_all.contiguous())
        ctx.owner.schedule_flush()
        return (None,) * 7


def _tbptt_backward(owner, bounds, dims, saved_state, unified, dd_all):
    """The backward launches of a TBPTT pass for d loss / d deltas = ``dd_all`` [T,B,1,N] (every chunk in the same three
    chunk launches when there are at most MAX_SPANS of them, then all encoder backward jobs); marks the packs dirty."""
    b, t_total, n, nchunks = dims
    actions_t, lactions_t, seeds, lstates, h0s, c0s, h_alls, c_alls, saveds, asaved, ssaved = saved_state
    lib = load()
    dev = actions_t.device
    dxlat_all = torch.empty_like(lactions_t)
    rows = max(b, CHUNK_ROWS)
    owner.chunk.ensure_rows(nchunks * rows)
    enc_rows = [min(ENCODER_ROWS, ls.shape[0] * b) for ls in lstates]
    owner.state_enc.ensure_rows(sum(enc_rows))
    owner.refresh_partials()
    dlsts = [torch.empty_like(ls) for ls in lstates]
    s_lat = owner.chunk.c.cs * owner.chunk.c.hq
    if nchunks <= MAX_SPANS:
        # every chunk in the same three launches (decoder backward, cell chains, dx + weight gradients): no forks
        h_all_u, c_all_u, saved_u = unified
        spans = (ChunkSpan * nchunks)()
        for c, (k0, k1) in enumerate(bounds):
            spans[c] = ChunkSpan(k0, k1, min(seeds[c].shape[0], k1 - k0), lstates[c].data_ptr(), h0s[c].data_ptr(),
                                 c0s[c].data_ptr(), 0 if c == 0 else s_lat, dlsts[c].data_ptr())
        rows_all = max(nchunks * b, min(nchunks * rows, t_total * b))
        owner.chunk.ensure_rows(rows_all)
        owner.refresh_partials()
        work = _chunk_workspace(owner.chunk, t_total, b, dev)
        _check(lib.sur_chunks_backward(_stream(), ctypes.byref(owner.chunk.c), nchunks, spans, _p(lactions_t), _p(h_all_u),
                                       _p(c_all_u), _p(dd_all), t_total, b, _p(dxlat_all), 0, rows_all, _p(saved_u), _p(work)))
    else:
        streams = _side_streams(owner, dev, nchunks)
        forks = []
        for c, (k0, k1) in enumerate(bounds):
            fork = _Fork(streams[c])
            with fork:
                work = _chunk_workspace(owner.chunk, k1 - k0, b, dev)
                _check(lib.sur_chunk_backward(_stream(), ctypes.byref(owner.chunk.c), _p(lactions_t[k0:k1]), _p(lstates[c]),
                                              _p(h0s[c]), _p(c0s[c]), 0 if c == 0 else s_lat, _p(h_alls[c]), _p(c_alls[c]),
                                              _p(dd_all[k0:k1]), None, None, None, k1 - k0, min(seeds[c].shape[0], k1 - k0), b,
                                              _p(dxlat_all[k0:k1]), _p(dlsts[c]), None, None, c * rows, rows, _p(saveds[c]),
                                              _p(work)))
                for t in (work, dlsts[c]):
                    t.record_stream(fork.stream)
            forks.append(fork)
        for fork in forks:
            fork.join()
    # every encoder backward of the step in launches of up to three jobs: all their workgroups are dispatched
    # together (as separate launches the long action-encoder job queued behind a state-encoder job), longest first
    jobs = [(owner.action_enc, actions_t, dxlat_all, t_total * b, 0, min(ENCODER_ROWS, owner.action_enc.c.rows), asaved)]
    row0 = 0
    for c in range(nchunks):
        jobs.append((owner.state_enc, seeds[c], dlsts[c], lstates[c].shape[0] * b, row0, enc_rows[c], ssaved[c]))
        row0 += enc_rows[c]
    for j0 in range(0, len(jobs), 3):
        _encoder_backward_multi(lib, jobs[j0:j0 + 3])
    for pack in owner.packs:
        pack.dirty = True


def fused_tbptt(surrogate, states, actions, tau, tbtt):
    """TBPTT forward of PDETrainingModule (training.py:71-98) on the fused kernels.  Returns
    (outputs [B,T,1,N], outdeltas [B,T,1,N], (H, C), time-major outdeltas [T,B,1,N]) with one action per step
    (the training layout)."""
    b, _, _, n = states.shape
    owner = packs_for(surrogate, n, b)
    d_all, out_all, h, c = _TBPTTFn.apply(states, actions, owner.anchor, owner, surrogate, tau, tbtt)
    return out_all.transpose(0, 1), d_all.transpose(0, 1), (h, c), d_all


PIPE_CHUNK_ROWS = 768   # partial rows (= workgroups of the pair-parallel backward kernels) of ONE chunk's backward branch


def fused_tbptt_train(surrogate, states, actions, tau, tbtt, delta, mean, stdv):
    """One whole TBPTT training pass WITHOUT autograd -- forward, delta loss, backward, gradient reduction (and the Adam
    step, when the packs have been lent the optimizer descriptors) -- with the chunks **pipelined**: TBPTT cuts the graph
    between chunks (training.py:71-98) and the delta loss is a plain sum over time steps (training.py:100-121), so the
    loss rows + backward kernels + encoder backward of chunk c are queued on a branch stream as soon as chunk c has been
    rolled out, and run while chunk c+1 is still going forward (its cell chains occupy B of the 256 CUs).  Only the last
    chunk's backward is left on the critical path, at about half the (step, sample) pairs per launch.  Under hipGraph
    capture the branches become parallel graph branches (each forked from the capturing stream itself: nested forks crash
    hipStreamEndCapture on ROCm 7).

    The gradient scale is fixed at d loss = 1: this is the captured training step (``GraphedTBPTTStep``); code that
    calls ``loss.backward()`` itself goes through ``fused_tbptt`` + ``fused_delta_loss``.
    Returns (outputs [B,T,1,N], outdeltas [B,T,1,N], (H, C), loss, hsteploss [T-1], stats [4], true deltas [B,T-1,1,N]),
    or None when the pass cannot be pipelined (a single chunk, inner forks disabled, no saved-activation records)."""
    b, t_total, _, n = actions.shape
    if not _INNER_FORKS or t_total <= tbtt or t_total < 2:
        return None
    owner = packs_for(surrogate, n, b)
    lib = load()
    dev = actions.device
    if not SAVE_ACTIVATIONS or min(lib.sur_encoder_saved_floats(ctypes.byref(owner.action_enc.c)),
                                   lib.sur_encoder_saved_floats(ctypes.byref(owner.state_enc.c))) <= 0:
        return None
    bounds = [(k0, min(k0 + tbtt, t_total)) for k0 in range(0, t_total, tbtt)]
    nchunks = len(bounds)
    if states.stride(3) != 1:
        states = states.contiguous()
    # partial-gradient rows of every concurrent launch: disjoint, and reserved before the first launch (growing a buffer
    # re-points the packs)
    chunk_rows = [max(b, min((k1 - k0) * b, PIPE_CHUNK_ROWS)) for k0, k1 in bounds]
    act_rows = [min(ENCODER_ROWS, (k1 - k0) * b) for k0, k1 in bounds]
    st_rows = [min(ENCODER_ROWS, (tau if c == 0 else 1) * b) for c in range(nchunks)]
    # row layout per pack: [ last chunk's rows | ONE fold row | chunk 0's rows | chunk 1's rows | ... ].  Each side branch
    # folds its own rows into the fold row when it is done (sur_fold_rows), so the flush at the end of the step reads the
    # last chunk's rows + 1 instead of every chunk's.
    def layout(rows):
        base, nxt = [0] * nchunks, rows[-1] + 1
        for c in range(nchunks - 1):
            base[c], nxt = nxt, nxt + rows[c]
        return base, rows[-1], nxt          # per-chunk row base, fold row, rows in total
    chunk_base, chunk_fold, chunk_total = layout(chunk_rows)
    act_base, act_fold, act_total = layout(act_rows)
    st_base, st_fold, st_total = layout(st_rows)
    owner.chunk.ensure_rows(chunk_total)
    owner.action_enc.ensure_rows(act_total)
    owner.state_enc.ensure_rows(st_total)
    owner.refresh_partials()
    new = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
    deltas, hstep, loss, stats = new(b, t_total - 1, 1, n), new(t_total - 1), new(), new(4)
    dd_all = new(t_total, b, 1, n)
    scratch = owner.loss_scratch.get(t_total)
    if scratch is None:
        scratch = (torch.empty(40 * t_total, device=dev, dtype=torch.float64), torch.zeros(1, device=dev, dtype=torch.int32))
        owner.loss_scratch[t_total] = scratch
    partial, ticket = scratch
    # ONE side stream carries everything that is off the critical path, in order: the action latents of the later chunks
    # (_tbptt_forward), then the backward branch of chunk 0, of chunk 1, ...  A captured graph then has exactly two
    # parallel node lists = two HIP streams; with three, two of them landed on the same hardware queue on some runs
    # (ROCm maps streams onto GPU_MAX_HW_QUEUES = 4 queues) and the later chunks' latents queued behind a whole backward
    # branch.  The backward kernels fill the device by themselves, so serialising the branches costs nothing.
    (side,) = _side_streams(owner, dev, 1)
    forks, keep = [], []

    def backward_of(c, st):
        k0, k1 = bounds[c]
        k = k1 - k0
        # the last chunk's loss rows are on the critical path: rows only there (the reduction to loss / statistics and the
        # integration of its predictions follow on the side stream, `finish_last`)
        loss_rows = lib.sur_tbptt_delta_loss_rows if c == nchunks - 1 else lib.sur_tbptt_delta_loss_range
        _check(loss_rows(_stream(), _p(states), states.stride(0), states.stride(1), _p(st.d_all), b, t_total, n, float(delta),
                         float(mean), float(stdv), _p(deltas), _p(dd_all), _p(hstep), _p(loss), _p(stats), _p(partial),
                         _p(ticket), k0, k1))
        if c == nchunks - 1:
            rows_done = torch.cuda.Event()
            rows_done.record(torch.cuda.current_stream(dev))
            tail.append(rows_done)
        dxlat = torch.empty_like(st.lactions_t[k0:k1])
        dlst = torch.empty_like(st.lstates[c])
        work = _chunk_workspace(owner.chunk, k, b, dev)
        _check(lib.sur_chunk_backward(_stream(), ctypes.byref(owner.chunk.c), _p(st.lactions_t[k0:k1]), _p(st.lstates[c]),
                                      _p(st.h0s[c]), _p(st.c0s[c]), 0 if c == 0 else st.s_lat, _p(st.h_alls[c]), _p(st.c_alls[c]),
                                      _p(dd_all[k0:k1]), None, None, None, k, min(st.seeds[c].shape[0], k), b, _p(dxlat),
                                      _p(dlst), None, None, chunk_base[c], chunk_rows[c], _p(st.saveds[c]), _p(work)))
        _encoder_backward_multi(lib, [
            (owner.action_enc, st.actions_t[k0:k1], dxlat, k * b, act_base[c], act_rows[c], st.asaved[k0 * b:k1 * b]),
            (owner.state_enc, st.seeds[c], dlst, st.lstates[c].shape[0] * b, st_base[c], st_rows[c], st.ssaved[c])])
        if c < nchunks - 1:
            three = ctypes.c_int * 3
            _check(lib.sur_fold_rows(_stream(), ctypes.byref(owner.state_enc.c), ctypes.byref(owner.action_enc.c),
                                     ctypes.byref(owner.chunk.c), three(st_base[c], act_base[c], chunk_base[c]),
                                     three(st_rows[c], act_rows[c], chunk_rows[c]), three(st_fold, act_fold, chunk_fold)))
        keep.extend((dxlat, dlst, work))

    pending = []     # (chunk, fork point) whose backward branch has not been issued yet
    tail = []        # the event after the last chunk's loss rows

    def finish_last(st):
        """Side stream, after every branch: integrate the last chunk's predictions, reduce the loss partial sums."""
        k0, k1 = bounds[-1]
        fork = _Fork(side, after=tail[0])
        with fork:
            _check(lib.sur_chunk_integrate(_stream(), ctypes.byref(owner.chunk.c), _p(st.seeds[-1]), _p(st.d_all[k0:k1]), k1 - k0,
                                           min(st.seeds[-1].shape[0], k1 - k0), b, _p(st.out_all[k0:k1])))
            _check(lib.sur_tbptt_delta_loss_finalize(_stream(), b, t_total, n, _p(hstep), _p(loss), _p(stats), _p(partial),
                                                     _p(ticket)))
        forks.append(fork)

    def after_chunk(c, st):
        # chunk c-1's branch forks where chunk c-1's rollout ended but is issued only now, after chunk c's own forward
        # launches: the critical path's nodes must be created before a fork point's other successors (see _Fork)
        for pc, point in pending:
            fork = _Fork(side, after=point)
            with fork:
                backward_of(pc, st)
            forks.append(fork)
        del pending[:]
        if c == nchunks - 1:
            backward_of(c, st)            # the last chunk's backward IS the critical path: it stays on the main stream
            return
        point = torch.cuda.Event()
        point.record(torch.cuda.current_stream(dev))
        pending.append((c, point))

    st = _tbptt_forward(states, actions, owner, surrogate, tau, tbtt, after_chunk=after_chunk, integrate_last=False)
    finish_last(st)
    for fork in forks:
        fork.join()
    for pack, extent in ((owner.state_enc, st_fold + 1), (owner.action_enc, act_fold + 1), (owner.chunk, chunk_fold + 1)):
        pack.dirty = True
        pack.c.rows = extent            # what the flush reduces and re-zeroes: everything beyond has been folded (and zeroed)
    try:
        owner.flush()
    finally:
        owner.refresh_partials()
    del keep[:]
    return (st.out_all.transpose(0, 1), st.d_all.transpose(0, 1), (st.h_alls[-1][-1], st.c_alls[-1][-1]), loss, hstep, stats,
            deltas)


def fused_tbptt_forward_loss(surrogate, states, actions, tau, tbtt, delta, mean, stdv):
    """Forward launches of a TBPTT pass + the one-launch delta loss, WITHOUT autograd: the first half of the split-graph
    step (``graph_step.GraphedAutogradStep``).  Returns (state for ``fused_tbptt_backward``, result tuple
    (outputs, outdeltas, (H, C), loss, hsteploss, stats, true deltas))."""
    b, t_total, _, n = actions.shape
    owner = packs_for(surrogate, n, b)
    dev = actions.device
    if states.stride(3) != 1:
        states = states.contiguous()
    st = _tbptt_forward(states, actions, owner, surrogate, tau, tbtt)
    new = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
    deltas, hstep, loss, stats = new(b, t_total - 1, 1, n), new(t_total - 1), new(), new(4)
    st.dd_all = new(t_total, b, 1, n)
    scratch = owner.loss_scratch.get(t_total)
    if scratch is None:
        scratch = (torch.empty(40 * t_total, device=dev, dtype=torch.float64), torch.zeros(1, device=dev, dtype=torch.int32))
        owner.loss_scratch[t_total] = scratch
    _check(load().sur_tbptt_delta_loss(_stream(), _p(states), states.stride(0), states.stride(1), _p(st.d_all), b, t_total, n,
                                       float(delta), float(mean), float(stdv), _p(deltas), _p(st.dd_all), _p(hstep), _p(loss),
                                       _p(stats), _p(scratch[0]), _p(scratch[1])))
    st.owner = owner
    return st, (st.out_all.transpose(0, 1), st.d_all.transpose(0, 1), (st.h_alls[-1][-1], st.c_alls[-1][-1]), loss, hstep, stats,
                deltas)


def fused_tbptt_backward(st, accumulate=False):
    """Second half of the split-graph step: the backward launches from ``st.dd_in`` (= ``st.dd_all`` times the incoming
    gradient, written by the caller; ``st.dd_all`` itself when absent) and the gradient reduction into the packs' flat
    buffers."""
    saved = (st.actions_t, st.lactions_t, st.seeds, st.lstates, st.h0s, st.c0s, st.h_alls, st.c_alls, st.saveds, st.asaved, st.ssaved)
    dd = getattr(st, "dd_in", None)      # the caller's (scaled) copy of d loss / d deltas, when it keeps st.dd_all pristine
    _tbptt_backward(st.owner, st.bounds, (st.b, st.t_total, st.n, st.nchunks), saved, (st.h_all_u, st.c_all_u, st.saved_u),
                    st.dd_all if dd is None else dd)
    st.owner.flush_into_flat(accumulate)
