"""Whole-network FNO kernels behind ``FNOAutoRegSurrogate.rollout`` (C ABI: include/spectral_hip.h ``fno_*``; kernels:
csrc/fno.hip) -- SURVEY 8(f) row f4, BASELINE configs[4].

The rollout contract is the reference's (pdecontrol/surrogates/surrogate.py:79-133): teacher forced on the given states,
free running afterwards, ``next = prev + delta * dscaling(model(prev, action))``.  Here a rollout of K steps is

  forward   ONE launch for all teacher-forced steps (n_given x B independent (step, sample) pairs), then one launch per
            free-running step (B pairs; the recurrence is the only sequential part);
  backward  the same launches in reverse (the gradient with respect to a step's starting state is handed to the step
            that predicted it), then ONE reduction of the per-pair parameter-gradient rows and ONE contraction of the
            saved spectra into the spectral weight gradients.

``_FNORolloutFn`` is a single autograd node per rollout, ``_FNOTBPTTFn`` one node for the training module's whole TBPTT pass
(all chunks; their backward chains run concurrently): parameters go in as inputs, their gradients come out of ``backward`` --
ordinary autograd, so optimizers, DDP hooks and hipGraph capture see nothing unusual.  There is no
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


def _launch_groups(n_given, K):
    """(first step, steps) per launch: all teacher-forced steps at once, then one free-running step at a time."""
    return [(0, n_given)] + [(k, 1) for k in range(n_given, K)]


def _forward_launches(lib, w, states, acts, n_given, cscale, cshift, deltas, outputs, pre, xspec, spec_pairs, pair0):
    """The forward launches of one rollout: ``states`` [B, S, 1, N] (the given states), ``acts`` [B, K, 1, N]; writes
    deltas / outputs [K, B, N] and, when ``pre`` is given, the saved tensors (``pre`` [K * B, ...]; the spectra go into the
    window [pair0, pair0 + K * B) of a buffer spanning ``spec_pairs`` pairs)."""
    B, K, N = acts.shape[0], acts.shape[1], acts.shape[-1]
    st = _stream()
    ast, asb = acts.stride(1), acts.stride(0)
    for k0, cnt in _launch_groups(n_given, K):
        if k0 < n_given:
            u_ptr, ust, usb = states.data_ptr(), states.stride(1), states.stride(0)
        else:                      # free running: the previous prediction
            u_ptr, ust, usb = outputs[k0 - 1].data_ptr(), 0, outputs.stride(1)
        _check(lib.fno_forward(st, ctypes.byref(w), WIDTH, MODES, LAYERS, N, B, cnt * B, ctypes.c_void_p(u_ptr), ust, usb,
                               ctypes.c_void_p(acts.data_ptr() + 4 * k0 * ast), ast, asb, cscale, cshift,
                               _ptr(deltas[k0]), _ptr(outputs[k0]), _ptr(None if pre is None else pre[k0 * B:]), _ptr(xspec),
                               spec_pairs, pair0 + k0 * B))


def _backward_launches(lib, w, states, acts, outputs, pre, n_given, cscale, g_deltas, gspec, spec_pairs, pair0, rows, gouts):
    """The backward launches of one rollout, last step first.  ``rows`` [K * B, width] and the spectra window receive the
    per-pair results; ``gouts``: K - n_given preallocated [B, N] buffers for the gradient a free-running step sends to the
    prediction it started from (it belongs to the step before it: free steps hand it on one by one, the teacher-forced
    launch receives it for its LAST step only -- true states need no gradient)."""
    B, K, N = acts.shape[0], acts.shape[1], acts.shape[-1]
    st = _stream()
    ast, asb = acts.stride(1), acts.stride(0)
    gout = None                     # d loss / d outputs[k0 - 1], produced by the launch of step k0
    for k0, cnt in reversed(_launch_groups(n_given, K)):
        free = k0 >= n_given
        if free:
            u_ptr, ust, usb = outputs[k0 - 1].data_ptr(), 0, outputs.stride(1)
        else:
            u_ptr, ust, usb = states.data_ptr(), states.stride(1), states.stride(0)
        new_gout = gouts[k0 - n_given] if free else None
        _check(lib.fno_backward(st, ctypes.byref(w), WIDTH, MODES, LAYERS, N, B, cnt * B, ctypes.c_void_p(u_ptr), ust, usb,
                                ctypes.c_void_p(acts.data_ptr() + 4 * k0 * ast), ast, asb, cscale,
                                _ptr(g_deltas[k0]), _ptr(gout), (0 if free else n_given - 1), _ptr(pre[k0 * B:]), _ptr(gspec),
                                spec_pairs, pair0 + k0 * B, _ptr(rows[k0 * B:]), _ptr(new_gout)))
        gout = new_gout


def _parameter_grads(lib, rows, xspec, gspec, pairs, needs):
    """One reduction of the per-pair rows + one contraction of the saved spectra -> the gradients in parameters_of order."""
    dev = rows.device
    st = _stream()
    width = rows.shape[1]
    flat = torch.empty(width, device=dev, dtype=torch.float32)
    _check(lib.fno_reduce_rows(st, _ptr(rows), pairs, _ptr(flat)))
    dwr = [torch.empty((WIDTH, WIDTH, MODES), device=dev, dtype=torch.float32) for _ in range(LAYERS)]
    dwi = [torch.empty((WIDTH, WIDTH, MODES), device=dev, dtype=torch.float32) for _ in range(LAYERS)]
    _check(lib.fno_spec_wgrad(st, _ptr(xspec), _ptr(gspec), pairs, (_p * 4)(*[t.data_ptr() for t in dwr]),
                              (_p * 4)(*[t.data_ptr() for t in dwi])))
    grads = [flat[0:64].view(32, 2), flat[64:96]]
    off = 96
    for layer in range(LAYERS):
        grads += [dwr[layer], dwi[layer], flat[off:off + 1024].view(32, 32), flat[off + 1024:off + 1056]]
        off += 1056
    grads += [flat[off:off + 1024].view(32, 32), flat[off + 1024:off + 1056], flat[off + 1056:off + 1088].view(1, 32),
              flat[off + 1088:off + 1089]]
    return [g if need else None for g, need in zip(grads, needs)]


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
        _forward_launches(lib, w, states, acts, n_given, cscale, cshift, deltas, outputs, pre, xspec, K * B, 0)
        ctx.need = need
        if need:
            ctx.save_for_backward(states, acts, outputs, pre, xspec, *params)
            ctx.meta = (B, K, N, n_given, float(cscale))
        ctx.mark_non_differentiable(outputs)
        del keep
        return deltas, outputs

    @staticmethod
    def backward(ctx, g_deltas, _g_outputs):
        if not ctx.need:
            return (None,) * len(ctx.needs_input_grad)
        lib = load()
        B, K, N, n_given, cscale = ctx.meta
        saved = ctx.saved_tensors
        states, acts, outputs, pre, xspec = saved[:5]
        params = saved[5:]
        dev = states.device
        w, keep = _weights_struct(params)
        g_deltas = g_deltas.contiguous()
        rows = torch.empty((K * B, lib.fno_row_width()), device=dev, dtype=torch.float32)
        gspec = torch.empty((LAYERS, 2 * MODES, K * B, WIDTH), device=dev, dtype=torch.float32)
        gouts = [torch.empty((B, N), device=dev, dtype=torch.float32) for _ in range(K - n_given)]
        _backward_launches(lib, w, states, acts, outputs, pre, n_given, cscale, g_deltas, gspec, K * B, 0, rows, gouts)
        out = _parameter_grads(lib, rows, xspec, gspec, K * B, ctx.needs_input_grad[5:])
        del keep
        return (None, None, None, None, None, *out)


class _FNOTBPTTFn(torch.autograd.Function):
    """The whole TBPTT pass of the training module (pdecontrol/surrogates/training.py:71-98 in the reference: chunks of
    ``tbtt`` steps, chunk 0 teacher forced on the first ``tau`` states, later chunks seeded by the previous chunk's last
    prediction with the gradient cut) as ONE autograd node: deltas / outputs [T, B, N] time major.  Forward: the chunks'
    launches in order (a chunk needs its predecessor's last prediction).  Backward: TBPTT cuts the graph between chunks, so
    their backward chains are independent -- each runs on its own stream (under hipGraph capture: parallel branches), 64 of
    the 256 CUs each at B = 64 --, then ONE row reduction and ONE spectra contraction for all chunks."""

    @staticmethod
    def forward(ctx, states, actions, tau, tbtt, cscale, cshift, *params):
        lib = load()
        states, actions = states.contiguous(), actions.contiguous()     # [B, T, 1, N] each
        B, T, N = actions.shape[0], actions.shape[1], actions.shape[-1]
        dev = states.device
        need = any(ctx.needs_input_grad[6:])
        w, keep = _weights_struct(params)
        deltas = torch.empty((T, B, N), device=dev, dtype=torch.float32)
        outputs = torch.empty((T, B, N), device=dev, dtype=torch.float32)
        pre = torch.empty((T * B, LAYERS, WIDTH, N), device=dev, dtype=torch.float32) if need else None
        xspec = torch.empty((LAYERS, 2 * MODES, T * B, WIDTH), device=dev, dtype=torch.float32) if need else None
        chunks = []                       # (k0, K, n_given, seed states [B, S, 1, N])
        for k0 in range(0, T, tbtt):
            K = min(tbtt, T - k0)
            if k0 == 0:
                seed, n_given = states[:, :min(tau, K)], min(tau, K)
            else:                         # the previous chunk's last prediction, as a [B, 1, 1, N] view of the time-major buffer
                seed, n_given = outputs[k0 - 1].view(B, 1, 1, N), 1
            chunks.append((k0, K, n_given, seed))
            _forward_launches(lib, w, seed, actions[:, k0:k0 + K], n_given, cscale, cshift, deltas[k0:], outputs[k0:],
                              None if pre is None else pre[k0 * B:], xspec, T * B, k0 * B)
        ctx.need = need
        if need:
            ctx.save_for_backward(states, actions, outputs, pre, xspec, *params)
            ctx.meta = (B, T, N, float(cscale), [(k0, K, ng) for k0, K, ng, _ in chunks], int(tau))
        ctx.mark_non_differentiable(outputs)
        del keep
        return deltas, outputs

    @staticmethod
    def backward(ctx, g_deltas, _g_outputs):
        if not ctx.need:
            return (None,) * len(ctx.needs_input_grad)
        from pdecontrol.surrogates import hipops
        lib = load()
        B, T, N, cscale, chunks, tau = ctx.meta
        saved = ctx.saved_tensors
        states, actions, outputs, pre, xspec = saved[:5]
        params = saved[5:]
        dev = states.device
        w, keep = _weights_struct(params)
        g_deltas = g_deltas.contiguous()
        # every buffer is allocated here, on the current stream, before the branches fork
        rows = torch.empty((T * B, lib.fno_row_width()), device=dev, dtype=torch.float32)
        gspec = torch.empty((LAYERS, 2 * MODES, T * B, WIDTH), device=dev, dtype=torch.float32)
        gouts = [[torch.empty((B, N), device=dev, dtype=torch.float32) for _ in range(K - ng)] for _, K, ng in chunks]
        streams = hipops.pooled_streams(dev, max(len(chunks) - 1, 0), "side")
        forks = []
        for c, (k0, K, ng) in enumerate(chunks):
            seed = states[:, :ng] if k0 == 0 else outputs[k0 - 1].view(B, 1, 1, N)
            args = (lib, w, seed, actions[:, k0:k0 + K], outputs[k0:], pre[k0 * B:], ng, cscale, g_deltas[k0:], gspec, T * B,
                    k0 * B, rows[k0 * B:], gouts[c])
            if c + 1 < len(chunks):       # every chunk but the last on a side stream; the last one on the current stream
                fork = hipops._Fork(streams[c])
                with fork:
                    _backward_launches(*args)
                forks.append(fork)
            else:
                _backward_launches(*args)
        for fork in forks:
            fork.join()
        out = _parameter_grads(lib, rows, xspec, gspec, T * B, ctx.needs_input_grad[6:])
        del keep
        return (None, None, None, None, None, None, *out)


def tbptt(surrogate, states, actions, tau, tbtt, grid):
    """(outputs [B, T, 1, N], deltas [B, T, 1, N], time-major deltas [T, B, 1, N]) of the training module's TBPTT pass on the
    whole-network kernels, or None when they do not cover this call.  ``grid(K)`` -> (times, targets) of a K-action chunk
    (the training module's ``_grid``): the action applied at every internal step follows the reference's integer path
    (``action_and_target_indices``: ``searchsorted`` over a floating-point ``arange`` -- at delta = 0.05 the last step of a
    10-step chunk re-uses action 8), exactly as ``rollout`` does chunk by chunk."""
    from pdecontrol.surrogates.surrogate import action_and_target_indices, take_steps
    model = surrogate.model
    if states.dtype != torch.float32 or states.dim() != 4 or states.shape[2] != 1 or actions.shape != states.shape:
        return None
    n, T = states.shape[-1], actions.shape[1]
    if not supported(model, n):
        return None
    aff = affine_of(surrogate.dscaling)
    if aff is None:
        return None
    gidx = []
    for k0 in range(0, T, tbtt):
        K = min(tbtt, T - k0)
        aidx, tidx = action_and_target_indices(*grid(K), surrogate.delta)
        aidx, tidx = aidx.tolist(), tidx.tolist()
        # one reported step per internal step (a trailing internal step nobody reports is dropped: it feeds nothing)
        if tidx != list(range(K)) or len(aidx) < K or min(aidx[:K]) < 0 or max(aidx[:K]) >= K:
            return None
        gidx += [k0 + j for j in aidx[:K]]
    acts = actions if gidx == list(range(T)) else take_steps(actions, gidx)
    scale, shift = aff
    d, o = _FNOTBPTTFn.apply(states, acts, int(tau), int(tbtt), float(surrogate.delta) * scale, float(surrogate.delta) * shift,
                             *parameters_of(model))
    return o.transpose(0, 1).unsqueeze(2), d.transpose(0, 1).unsqueeze(2), d.unsqueeze(2)


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
