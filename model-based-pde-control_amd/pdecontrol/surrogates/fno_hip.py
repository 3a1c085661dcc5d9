"""Whole-network FNO kernels behind ``FNOAutoRegSurrogate.rollout`` (C ABI: include/spectral_hip.h ``fno_*``; kernels:
csrc/fno.hip) -- SURVEY 8(f) row f4, BASELINE configs[4].

The rollout contract is the reference's (pdecontrol/surrogates/surrogate.py:79-133): teacher forced on the given states,
free running afterwards, ``next = prev + delta * dscaling(model(prev, action))``.  Here a rollout of K steps is

  forward   ONE launch for all teacher-forced steps (n_given x B independent (step, sample) pairs), then one launch per
            free-running step (B pairs; the recurrence is the only sequential part);
  backward  the same launches in reverse (the gradient with respect to a step's starting state is handed to the step
            that predicted it), then ONE reduction of the per-pair parameter-gradient rows and ONE contraction of the
            saved spectra into the spectral weight gradients.

``_FNORolloutFn`` is a single autograd node per rollout: parameters go in as inputs, their gradients come out of
``backward`` -- ordinary autograd, so optimizers, DDP hooks and hipGraph capture see nothing unusual.  There is no
reference-side pin for any of this (the reference has no FNO): the kernels are pinned against the torch spelling
(tests/test_fno.py).
"""
import ctypes

import torch

from pdecontrol.surrogates import spectral

_p, _i, _l, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
WIDTH, MODES, LAYERS = 32, 16, 4


class FnoWeights(ctypes.Structure):
    _fields_ = [("lift_w", _p), ("lift_b", _p), ("spec_wr", _p * 4), ("spec_wi", _p * 4), ("pw_w", _p * 4), ("pw_b", _p * 4),
                ("p1_w", _p), ("p1_b", _p), ("p2_w", _p), ("p2_b", _p)]


_WP = ctypes.POINTER(FnoWeights)
SYMBOLS = (
    ("fno_forward", _i, [_p, _WP, _i, _i, _i, _i, _i, _i, _p, _l, _l, _p, _l, _l, _f, _f, _p, _p, _p, _p, _i, _i]),
    ("fno_backward", _i, [_p, _WP, _i, _i, _i, _i, _i, _i, _p, _l, _l, _p, _l, _l, _f, _p, _p, _i, _p, _p, _i, _i, _p, _p]),
    ("fno_row_width", _i, []),
    ("fno_reduce_rows", _i, [_p, _p, _i, _p]),
    ("fno_spec_wgrad", _i, [_p, _p, _p, _i, _p * 4, _p * 4]),
    ("fno_last_error", ctypes.c_char_p, []),
)
_lib = None


class FnoHipError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is None:
        lib = spectral.load()           # same shared library; raises when it has not been built
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise FnoHipError(f"libspectral_hip (fno) error {rc}: {load().fno_last_error().decode(errors='replace')}")


def _stream():
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def parameters_of(model):
    """The FNO1d parameters in the kernels' order (also the order of ``_FNORolloutFn``'s parameter inputs)."""
    ps = [model.lift.weight, model.lift.bias]
    for spec, pw in zip(model.spectral, model.pointwise):
        ps += [spec.weight_real, spec.weight_imag, pw.weight, pw.bias]
    ps += [model.project[0].weight, model.project[0].bias, model.project[2].weight, model.project[2].bias]
    return ps


def supported(model, n):
    """True when ``model`` is the FNO1d geometry the whole-network kernels are built for."""
    try:
        return (len(model.spectral) == LAYERS and model.lift.weight.shape == (WIDTH, 2)
                and all(s.in_channels == WIDTH and s.out_channels == WIDTH and s.modes == MODES for s in model.spectral)
                and isinstance(model.activation, torch.nn.GELU) and getattr(model.activation, "approximate", "none") == "none"
                and 64 <= n <= 512 and (n & (n - 1)) == 0 and model.lift.weight.dtype == torch.float32)
    except AttributeError:
        return False


def _weights_struct(params):
    w = FnoWeights()
    ps = [p.detach() if p.is_contiguous() else p.detach().contiguous() for p in params]
    w.lift_w, w.lift_b = ps[0].data_ptr(), ps[1].data_ptr()
    for layer in range(LAYERS):
        wr, wi, pw, pb = ps[2 + 4 * layer: 6 + 4 * layer]
        w.spec_wr[layer], w.spec_wi[layer], w.pw_w[layer], w.pw_b[layer] = wr.data_ptr(), wi.data_ptr(), pw.data_ptr(), pb.data_ptr()
    base = 2 + 4 * LAYERS
    w.p1_w, w.p1_b, w.p2_w, w.p2_b = (ps[base + k].data_ptr() for k in range(4))
    return w, ps      # ps keeps temporaries alive


def affine_of(dscaling):
    """(scale, shift) with dscaling(x) == x * scale + shift -- identity or the inverse of a scalar ``Normalize`` (what the
    controller fits, mbrl.py:168) -- or None when the transform is not of that form."""
    from pdecontrol.surrogates import hipops
    try:
        mul, add = hipops._dscale_constants(dscaling)
    except hipops.SurrogateHipError:
        return None
    return float(mul), float(add)


class _FNORolloutFn(torch.autograd.Function):
    """deltas [K, B, N], outputs [K, B, N] (time major) of a K-step rollout; see the module docstring."""

    @staticmethod
    def forward(ctx, states, acts, n_given, cscale, cshift, *params):
        lib = load()
        states, acts = states.contiguous(), acts.contiguous()          # [B, S, 1, N], [B, K, 1, N]
        B, K, N = acts.shape[0], acts.shape[1], acts.shape[-1]
        n_given = min(int(n_given), K)
        dev = states.device
        need = any(ctx.needs_input_grad[5:])      # (grad mode is off inside forward: ask the node, not torch.is_grad_enabled)
        w, keep = _weights_struct(params)
        deltas = torch.empty((K, B, N), device=dev, dtype=torch.float32)
        outputs = torch.empty((K, B, N), device=dev, dtype=torch.float32)
        pre = torch.empty((K * B, LAYERS, WIDTH, N), device=dev, dtype=torch.float32) if need else None
        xspec = torch.empty((LAYERS, 2 * MODES, K * B, WIDTH), device=dev, dtype=torch.float32) if need else None
        # launch groups (first step, steps): all teacher-forced steps at once, then one free-running step at a time
        groups = [(0, n_given)] + [(k, 1) for k in range(n_given, K)]
        st = _stream()
        ast, asb = acts.stride(1), acts.stride(0)
        for k0, cnt in groups:
            if k0 < n_given:
                u_ptr, ust, usb = states.data_ptr(), states.stride(1), states.stride(0)
            else:                      # free running: the previous prediction
                u_ptr, ust, usb = outputs[k0 - 1].data_ptr(), 0, outputs.stride(1)
            _check(lib.fno_forward(st, ctypes.byref(w), WIDTH, MODES, LAYERS, N, B, cnt * B, ctypes.c_void_p(u_ptr), ust, usb,
                                   ctypes.c_void_p(acts.data_ptr() + 4 * k0 * ast), ast, asb, cscale, cshift,
                                   _ptr(deltas[k0]), _ptr(outputs[k0]), _ptr(pre[k0 * B:] if need else None), _ptr(xspec),
                                   K * B, k0 * B))
        ctx.need = need
        if need:
            ctx.save_for_backward(states, acts, outputs, pre, xspec, *params)
            ctx.meta = (B, K, N, n_given, float(cscale), groups)
        ctx.mark_non_differentiable(outputs)
        del keep
        return deltas, outputs

    @staticmethod
    def backward(ctx, g_deltas, _g_outputs):
        if not ctx.need:
            return (None,) * len(ctx.needs_input_grad)
        lib = load()
        B, K, N, n_given, cscale, groups = ctx.meta
        saved = ctx.saved_tensors
        states, acts, outputs, pre, xspec = saved[:5]
        params = saved[5:]
        dev = states.device
        w, keep = _weights_struct(params)
        g_deltas = g_deltas.contiguous()
        width = lib.fno_row_width()
        rows = torch.empty((K * B, width), device=dev, dtype=torch.float32)
        gspec = torch.empty((LAYERS, 2 * MODES, K * B, WIDTH), device=dev, dtype=torch.float32)
        st = _stream()
        ast, asb = acts.stride(1), acts.stride(0)
        gout = None                     # d loss / d outputs[k0 - 1], produced by the launch of step k0
        for k0, cnt in reversed(groups):
            free = k0 >= n_given
            if free:
                u_ptr, ust, usb = outputs[k0 - 1].data_ptr(), 0, outputs.stride(1)
            else:
                u_ptr, ust, usb = states.data_ptr(), states.stride(1), states.stride(0)
            # the gradient a free step sends to its starting state belongs to the step before it: free steps hand it on
            # one by one; the teacher-forced launch receives it for its LAST step only (true states need no gradient)
            new_gout = torch.empty((B, N), device=dev, dtype=torch.float32) if free else None
            _check(lib.fno_backward(st, ctypes.byref(w), WIDTH, MODES, LAYERS, N, B, cnt * B, ctypes.c_void_p(u_ptr), ust, usb,
                                    ctypes.c_void_p(acts.data_ptr() + 4 * k0 * ast), ast, asb, cscale,
                                    _ptr(g_deltas[k0]), _ptr(gout), (0 if free else n_given - 1), _ptr(pre[k0 * B:]), _ptr(gspec),
                                    K * B, k0 * B, _ptr(rows[k0 * B:]), _ptr(new_gout)))
            gout = new_gout
        # parameter gradients: one reduction of the rows, one contraction of the spectra
        flat = torch.empty(width, device=dev, dtype=torch.float32)
        _check(lib.fno_reduce_rows(st, _ptr(rows), K * B, _ptr(flat)))
        dwr = [torch.empty((WIDTH, WIDTH, MODES), device=dev, dtype=torch.float32) for _ in range(LAYERS)]
        dwi = [torch.empty((WIDTH, WIDTH, MODES), device=dev, dtype=torch.float32) for _ in range(LAYERS)]
        _check(lib.fno_spec_wgrad(st, _ptr(xspec), _ptr(gspec), K * B, (_p * 4)(*[t.data_ptr() for t in dwr]),
                                  (_p * 4)(*[t.data_ptr() for t in dwi])))
        grads = [flat[0:64].view(32, 2), flat[64:96]]
        off = 96
        for layer in range(LAYERS):
            grads += [dwr[layer], dwi[layer], flat[off:off + 1024].view(32, 32), flat[off + 1024:off + 1056]]
            off += 1056
        grads += [flat[off:off + 1024].view(32, 32), flat[off + 1024:off + 1056], flat[off + 1056:off + 1088].view(1, 32),
                  flat[off + 1088:off + 1089]]
        del keep
        out = [g if need else None for g, need in zip(grads, ctx.needs_input_grad[5:])]
        return (None, None, None, None, None, *out)


def rollout(model, states, actions, n_given, delta, dscaling):
    """(deltas, outputs) [B, K, 1, N] of the K = actions.size(1)-step rollout, or None when the kernels do not cover this
    call (other geometry, non-affine dscaling, non-fp32): the caller then runs the per-operator path."""
    if states.dtype != torch.float32 or states.dim() != 4 or states.shape[2] != 1 or actions.shape[2] != 1:
        return None
    n = states.shape[-1]
    if not supported(model, n):
        return None
    aff = affine_of(dscaling)
    if aff is None:
        return None
    scale, shift = aff
    d, o = _FNORolloutFn.apply(states, actions, int(n_given), float(delta) * scale, float(delta) * shift, *parameters_of(model))
    return d.transpose(0, 1).unsqueeze(2), o.transpose(0, 1).unsqueeze(2)
