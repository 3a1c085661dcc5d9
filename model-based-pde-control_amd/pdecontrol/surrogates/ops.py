"""Functional layer between the surrogate modules and the kernels that execute them.

Every block of the surrogate goes through one of the functions below.  On CPU tensors they are
plain torch (this is also the fp32 torch reference the HIP kernels are tested against); on CUDA
tensors they use the hand-written gfx950 kernels of libsurrogate_hip.so when ``fused`` is enabled
(``enable_fused(True)`` -- it raises if the library is missing, there is no silent fallback once
fusion has been requested).
"""
import os

import torch

# PDECONTROL_FUSED=1 selects the fused kernels without touching the caller's code (the reference's script.py runs
# unmodified); the library is loaded at the first CUDA tensor and its absence raises there.
_FUSED = {"enabled": os.environ.get("PDECONTROL_FUSED", "0") == "1", "lib": None}


def enable_fused(flag=True):
    """Switch the CUDA code path to the fused HIP kernels (loads the library; raises if absent)."""
    if flag:
        from pdecontrol.surrogates import hipops
        _FUSED["lib"] = hipops.load()
    _FUSED["enabled"] = bool(flag)


def fused_enabled():
    return _FUSED["enabled"]


def use_fused(tensor):
    """True when the fused HIP rollout (hipops.fused_rollout) should handle this tensor."""
    if not (_FUSED["enabled"] and tensor.is_cuda):
        return False
    if _FUSED["lib"] is None:
        from pdecontrol.surrogates import hipops
        _FUSED["lib"] = hipops.load()
    return True


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
