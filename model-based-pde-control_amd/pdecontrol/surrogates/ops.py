"""Functional layer between the surrogate modules and the kernels that execute them.

Every block of the surrogate goes through one of the functions below.  On CPU tensors they are
plain torch (this is also the fp32 torch reference the HIP kernels are tested against).  On CUDA
tensors the hand-written gfx950 kernels of libsurrogate_hip.so are THE path for the architecture they implement
(``hipops.fused_supported``: the reference's KSAutoRegConvolutionalLSTM layout, at the grid widths whose LayerNorm rows
the kernels reduce -- N = 64, 128, 256, ``hipops.geometry_unsupported``; any other width is announced once and runs on torch
kernels like an uncovered architecture): they are on by default and the library's absence raises at the first CUDA tensor.  Every call site asks ``use_fused_for(surrogate, tensor)``: an
architecture the kernels do not cover -- the reference also ships ``KSAutoRegFullyConnectedLSTM`` and trains it on the
GPU as it is -- runs the plain PyTorch-ROCm path, announced ONCE per class through ``logging`` (never silently, and
never the other way round: a covered architecture cannot end up on torch kernels without the explicit opt-out).  The
plain torch-on-GPU path also exists as an explicit opt-out (``enable_fused(False)``, the
``fused(False)`` context manager or ``PDECONTROL_FUSED=0``): it is the same-device cross-check of the
parity tests and the "hipGraph over torch kernels" leg of the benchmark.
"""
import logging
import os

import torch

_LOG = logging.getLogger("pdecontrol.surrogates")
_NOTIFIED = set()
_DEFAULT = os.environ.get("PDECONTROL_FUSED", "1") != "0"
_FUSED = {"enabled": _DEFAULT, "lib": None}


def enable_fused(flag=True):
    """Select the CUDA code path: True = fused HIP kernels (loads the library; raises if absent)."""
    if flag:
        from pdecontrol.surrogates import hipops
        _FUSED["lib"] = hipops.load()
    _FUSED["enabled"] = bool(flag)


def reset_fused():
    """Back to the process default (fused unless PDECONTROL_FUSED=0)."""
    _FUSED["enabled"] = _DEFAULT


def fused_enabled():
    return _FUSED["enabled"]


class fused:
    """``with ops.fused(False): ...`` -- run a block on the other CUDA path and restore the previous choice."""

    def __init__(self, flag):
        self.flag = bool(flag)

    def __enter__(self):
        self.prev = _FUSED["enabled"]
        enable_fused(self.flag)
        return self

    def __exit__(self, *exc):
        _FUSED["enabled"] = self.prev


def use_fused(tensor):
    """True when the fused HIP kernels handle this tensor (any CUDA tensor unless opted out)."""
    if not (_FUSED["enabled"] and tensor.is_cuda):
        return False
    if _FUSED["lib"] is None:
        from pdecontrol.surrogates import hipops
        _FUSED["lib"] = hipops.load()   # raises when the library has not been built: no silent fallback
    return True


def use_fused_for(surrogate, tensor):
    """``use_fused(tensor)`` for a surrogate the fused kernels implement; for any other architecture False, with one
    logged notice per class (it then runs on PyTorch-ROCm kernels, as in the reference)."""
    if not (_FUSED["enabled"] and tensor.is_cuda):
        return False
    from pdecontrol.surrogates import hipops
    if hipops.fused_supported(surrogate):
        if not use_fused(tensor):
            return False
        reason = hipops.geometry_unsupported(surrogate, tensor.shape[-1])
        if reason is None:
            return True
        if reason not in _NOTIFIED:
            _NOTIFIED.add(reason)
            _LOG.warning("the fused HIP kernels do not implement this grid width (%s): it runs on plain PyTorch-ROCm kernels", reason)
        return False
    name = type(surrogate).__name__ + "/" + type(getattr(surrogate, "transition_model", None)).__name__
    if name not in _NOTIFIED:
        _NOTIFIED.add(name)
        _LOG.warning("%s is not the architecture the fused HIP kernels implement (KSAutoRegConvolutionalLSTM layout): "
                     "it runs on plain PyTorch-ROCm kernels", name)
    return False


def require_plain_path(tensor, what):
    """Surrogates the fused kernels do not implement must not run on MIOpen by accident."""
    if tensor.is_cuda and _FUSED["enabled"]:
        from pdecontrol.surrogates.hipops import SurrogateHipError
        raise SurrogateHipError(f"{what} has no fused HIP implementation; to run it on plain PyTorch-ROCm kernels opt out "
                                f"explicitly with pdecontrol.surrogates.ops.enable_fused(False) or PDECONTROL_FUSED=0")


def conv_act_norm(x, conv, activation, layernorm):
    """layernorm(activation(conv(x))) for nn.Conv1d (circular) / nn.ConvTranspose1d modules."""
    y = activation(conv(x))
    return y if layernorm is None else layernorm(y)


def lstm_cell(x, h, c, cell):
    """Convolutional LSTM cell (transition.py CNNLSTMCell.forward): returns (h', c')."""
    gi = cell.Wxi(x) + cell.Whi(h)
    gf = cell.Wxf(x) + cell.Whf(h)
    gc = cell.Wxc(x) + cell.Whc(h)
    go = cell.Wxo(x) + cell.Who(h)
    c_new = torch.sigmoid(gf) * c + torch.sigmoid(gi) * torch.tanh(gc)
    h_new = torch.sigmoid(go) * torch.tanh(c_new)
    return h_new, c_new
