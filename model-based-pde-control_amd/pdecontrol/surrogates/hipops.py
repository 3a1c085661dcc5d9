"""ctypes binding + autograd glue of libsurrogate_hip.so (C ABI: include/surrogate_hip.h).

``fused_rollout`` is the GPU implementation of ``AutoRegPDESurrogate.rollout`` for the
``KSAutoRegConvolutionalLSTM`` family: two encoder launches (all given states, all actions) and one
launch per time step (ConvLSTM cell + decoder + integration), each with a matching backward
launch.  Parameter gradients are accumulated by the kernels directly into ``param.grad`` (fp32
atomics), so the autograd graph only carries activations; a zero-dim ``anchor`` tensor makes the
custom Functions differentiable even when their data inputs are not.

There is no fallback in here: if the library is missing, ``load()`` raises.
"""
import ctypes
import os

import torch
from torch import nn

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.abspath(os.path.join(_HERE, "..", "..", "lib", "libsurrogate_hip.so"))

RB_NPARAM = 9
ST_NPARAM = 26
_fp = ctypes.c_void_p


class EncoderParams(ctypes.Structure):
    _fields_ = [("w", _fp * (3 * RB_NPARAM)), ("g", _fp * (3 * RB_NPARAM)), ("c", ctypes.c_int * 4),
                ("stride", ctypes.c_int * 3), ("n", ctypes.c_int)]


class StepParams(ctypes.Structure):
    _fields_ = [("w", _fp * ST_NPARAM), ("g", _fp * ST_NPARAM), ("ca", ctypes.c_int), ("cs", ctypes.c_int),
                ("hq", ctypes.c_int), ("c_mid", ctypes.c_int), ("delta", ctypes.c_float), ("mul", ctypes.c_float),
                ("add", ctypes.c_float)]


SYMBOLS = (
    ("sur_encoder_forward", [_fp, ctypes.POINTER(EncoderParams), _fp, ctypes.c_int, _fp]),
    ("sur_encoder_backward", [_fp, ctypes.POINTER(EncoderParams), _fp, _fp, ctypes.c_int, _fp]),
    ("sur_step_forward", [_fp, ctypes.POINTER(StepParams), _fp, _fp, _fp, _fp, ctypes.c_int, _fp, _fp, _fp, _fp]),
    ("sur_step_backward", [_fp, ctypes.POINTER(StepParams), _fp, _fp, _fp, _fp, _fp, _fp, _fp, ctypes.c_int, _fp, _fp,
                           _fp, _fp]),
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
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


# ---------------------------------------------------------------------------------------------
# parameter packs
# ---------------------------------------------------------------------------------------------
def _grad_of(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


class _Pack:
    """Pointers to the weights / gradient accumulators of one module, in the kernel's order."""

    def __init__(self, params, cstruct):
        self.params, self.c = params, cstruct
        self.refresh()

    def refresh(self):
        for i, p in enumerate(self.params):
            assert p.is_cuda and p.is_contiguous() and p.dtype == torch.float32
            self.c.w[i] = p.data_ptr()
            self.c.g[i] = _grad_of(p).data_ptr() if p.requires_grad else None


def _encoder_pack(convnet, n):
    from pdecontrol.surrogates.models.cnn import ResidualBlock
    blocks = [getattr(convnet, name) for name in convnet.layers]
    if len(blocks) != 3 or not all(isinstance(b, ResidualBlock) for b in blocks):
        raise SurrogateHipError("fused encoder expects three ResidualBlocks")
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
    return _Pack(params, c)


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


def _step_pack(surrogate):
    from pdecontrol.surrogates.models.cnn import ConvBlock, DeConvolutionBlock
    from pdecontrol.surrogates.transition import CNNLSTMTransitionModel
    tm = surrogate.transition_model
    if not isinstance(tm, CNNLSTMTransitionModel):
        raise SurrogateHipError("fused step expects a CNNLSTMTransitionModel")
    cell, dec = tm.cnnlstmcell, surrogate.state_decoder.model
    blocks = [getattr(dec, name) for name in dec.layers]
    ok = (len(blocks) == 4 and isinstance(blocks[0], DeConvolutionBlock) and isinstance(blocks[1], DeConvolutionBlock)
          and isinstance(blocks[2], ConvBlock) and isinstance(blocks[3], ConvBlock)
          and blocks[2].convolution.kernel_size == (7,) and blocks[3].convolution.kernel_size == (5,)
          and isinstance(blocks[3].activation, nn.Identity) and blocks[3].layernorm is None
          and all(isinstance(b.activation, nn.SiLU) for b in blocks[:3]))
    if not ok:
        raise SurrogateHipError("fused step expects the KSAutoRegConvolutionalLSTM decoder layout")
    params = []
    for gate in "ifco":
        wx, wh = getattr(cell, f"Wx{gate}"), getattr(cell, f"Wh{gate}")
        params += [wx.weight, wx.bias, wh.weight]
    d0, d1, c2, c3 = blocks
    params += [d0.deconvolution.weight, d0.deconvolution.bias, d0.layernorm.weight, d0.layernorm.bias,
               d1.deconvolution.weight, d1.deconvolution.bias, d1.layernorm.weight, d1.layernorm.bias,
               c2.convolution.weight, c2.convolution.bias, c2.layernorm.weight, c2.layernorm.bias,
               c3.convolution.weight, c3.convolution.bias]
    c = StepParams()
    c.ca, c.cs, c.hq = cell.in_channels, cell.out_channels, tm.ssize
    c.c_mid = d1.deconvolution.out_channels
    c.delta = float(surrogate.delta)
    c.mul, c.add = _dscale_constants(surrogate.dscaling)
    return _Pack(params, c)


class FusedPacks:
    def __init__(self, surrogate, n):
        load()
        self.n = n
        self.state_enc = _encoder_pack(surrogate.state_encoder.model, n)
        self.action_enc = _encoder_pack(surrogate.action_encoder.model, n)
        self.step = _step_pack(surrogate)
        dev = self.step.params[0].device
        self.anchor = torch.zeros((), device=dev, requires_grad=True)
        self.key = self._key(surrogate, n)

    @staticmethod
    def _key(surrogate, n):
        p = next(surrogate.parameters())
        return (p.data_ptr(), n)

    def refresh(self, surrogate):
        for pack in (self.state_enc, self.action_enc, self.step):
            pack.refresh()
        self.step.c.mul, self.step.c.add = _dscale_constants(surrogate.dscaling)


def packs_for(surrogate, n):
    packs = getattr(surrogate, "_fused_packs", None)
    if packs is None or packs.key != FusedPacks._key(surrogate, n):
        packs = FusedPacks(surrogate, n)
        object.__setattr__(surrogate, "_fused_packs", packs)
    else:
        packs.refresh(surrogate)
    return packs


# ---------------------------------------------------------------------------------------------
# autograd Functions
# ---------------------------------------------------------------------------------------------
class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, pack):
        x = x.contiguous()
        m, n = x.shape[0], pack.c.n
        h = n
        for s in pack.c.stride:
            h //= s
        z = torch.empty((m, pack.c.c[3], h), device=x.device, dtype=torch.float32)
        _check(load().sur_encoder_forward(_stream(), ctypes.byref(pack.c), _p(x), m, _p(z)))
        ctx.save_for_backward(x)
        ctx.pack, ctx.need_dx = pack, x.requires_grad
        return z

    @staticmethod
    def backward(ctx, dz):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x) if ctx.need_dx else None
        _check(load().sur_encoder_backward(_stream(), ctypes.byref(ctx.pack.c), _p(x), _p(dz.contiguous()), x.shape[0],
                                           _p(dx)))
        return dx, None, None


class _StepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xlat, h_in, c_prev, base, anchor, pack):
        xlat, h_in, c_prev, base = xlat.contiguous(), h_in.contiguous(), c_prev.contiguous(), base.contiguous()
        b = xlat.shape[0]
        h_out, c_out = torch.empty_like(h_in), torch.empty_like(c_prev)
        d_out, out = torch.empty_like(base), torch.empty_like(base)
        _check(load().sur_step_forward(_stream(), ctypes.byref(pack.c), _p(xlat), _p(h_in), _p(c_prev), _p(base), b,
                                       _p(h_out), _p(c_out), _p(d_out), _p(out)))
        ctx.save_for_backward(xlat, h_in, c_prev)
        ctx.pack = pack
        ctx.needs = (xlat.requires_grad, h_in.requires_grad, c_prev.requires_grad, base.requires_grad)
        ctx.set_materialize_grads(False)
        return h_out, c_out, d_out, out

    @staticmethod
    def backward(ctx, dh, dc, dd, dout):
        xlat, h_in, c_prev = ctx.saved_tensors
        nx, nh, nc, nb = ctx.needs
        cont = lambda t: None if t is None else t.contiguous()
        dh, dc, dd, dout = cont(dh), cont(dc), cont(dd), cont(dout)
        dxlat = torch.empty_like(xlat) if nx else None
        dh_in = torch.empty_like(h_in) if nh else None
        dc_prev = torch.empty_like(c_prev) if nc else None
        dbase = torch.empty_like(dout) if (nb and dout is not None) else None
        _check(load().sur_step_backward(_stream(), ctypes.byref(ctx.pack.c), _p(xlat), _p(h_in), _p(c_prev), _p(dd),
                                        _p(dout), _p(dh), _p(dc), xlat.shape[0], _p(dxlat), _p(dh_in), _p(dc_prev),
                                        _p(dbase)))
        return dxlat, dh_in, dc_prev, dbase, None, None


def encode(x, pack, anchor):
    """[M, C0, N] -> [M, C3, N/4] through the fused 3-block residual encoder."""
    return _EncoderFn.apply(x, anchor, pack)


def rollout_step(xlat, h_in, c_prev, base, pack, anchor):
    return _StepFn.apply(xlat, h_in, c_prev, base, anchor, pack)


def fused_rollout(surrogate, states, actions, times, targets, hidden):
    """GPU rollout of AutoRegPDESurrogate (same outputs as surrogate.py:79-133; ``inlatents`` is not
    produced).  states [B,S,1,N], actions [B,A,1,N]."""
    from pdecontrol.mbrl.types import ModelRollout
    from pdecontrol.surrogates.surrogate import action_and_target_indices, take_steps
    b, s_given, _, n = states.shape
    packs = packs_for(surrogate, n)
    a_steps = actions.shape[1]
    hq, cs, ca = packs.step.c.hq, packs.step.c.cs, packs.step.c.ca
    # time-major so that every per-step slice is contiguous
    states_t = states.transpose(0, 1).contiguous()                          # [S, B, 1, N]
    lstates_t = encode(states_t.reshape(s_given * b, 1, n), packs.state_enc, packs.anchor).reshape(s_given, b, cs, hq)
    aidx, tidx = action_and_target_indices(times, targets, surrogate.delta)
    actions_t = take_steps(actions, aidx.tolist()).transpose(0, 1).contiguous()   # [K, B, 1, N]
    n_steps = actions_t.shape[0]
    lactions_t = encode(actions_t.reshape(n_steps * b, 1, n), packs.action_enc, packs.anchor).reshape(n_steps, b, ca, hq)
    if hidden is None:
        tm = surrogate.transition_model
        hidden = (tm.H0.unsqueeze(0).expand(b, -1, -1).contiguous(), tm.C0.unsqueeze(0).expand(b, -1, -1).contiguous())
    H, C = hidden
    outs, deltas, latents = [], [], []
    output = states_t[0]
    for k in range(n_steps):
        if k < s_given:
            h_in, base = lstates_t[k], states_t[k]
        else:
            h_in, base = H, output
        H, C, d, output = rollout_step(lactions_t[k], h_in, C, base, packs.step, packs.anchor)
        outs.append(output)
        deltas.append(d)
        latents.append(H)
    pick = tidx.tolist()
    gather = lambda seq, shape: take_steps(torch.stack(seq, dim=1).reshape(shape), pick)
    return ModelRollout(inlatents=None, outlatents=gather(latents, (b, n_steps, cs, hq)),
                        deltas=gather(deltas, (b, n_steps, 1, n)), outputs=gather(outs, (b, n_steps, 1, n)),
                        hidden=(H, C))
