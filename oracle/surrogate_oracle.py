"""Functional CPU restatement of the reference's surrogate TBPTT step.  TEST INFRASTRUCTURE ONLY.

A second, module-free statement (pure ``torch.nn.functional`` on a dict of weights, fp32, CPU) of
the arithmetic in the reference's
  pdecontrol/surrogates/models/cnn.py:126-145   ResidualBlock.forward
  pdecontrol/surrogates/models/cnn.py:35-41,64-70  ConvBlock / DeConvolutionBlock.forward
  pdecontrol/surrogates/transition.py:218-226   CNNLSTMCell.forward
  pdecontrol/surrogates/transition.py:261-296   teacherforcing / transition
  pdecontrol/surrogates/surrogate.py:79-133     AutoRegPDESurrogate.rollout
  pdecontrol/surrogates/training.py:64-112      TBPTT chunking, detach, delta-mode MSE loss
for the ``KSAutoRegConvolutionalLSTM`` architecture (pdecontrol/architectures/autoreg.py:44-101).

Parity pin: tests/test_oracle_surrogate.py checks it against tests/golden/surrogate_golden.npz
(state_dict, inputs, loss, outputs, per-parameter gradients captured from the reference by
oracle/gen_golden.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.
"""
import torch
import torch.nn.functional as F


def load_weights(npz, prefix="sd/", requires_grad=True):
    """dict name -> fp32 tensor from the golden state_dict arrays."""
    w = {}
    for k in npz.files:
        if k.startswith(prefix):
            t = torch.from_numpy(npz[k]).clone()
            name = k[len(prefix):]
            if requires_grad and not name.endswith((".H0", ".C0")):
                t.requires_grad_(True)
            w[name] = t
    return w


def _circ_conv(x, weight, bias=None, stride=1):
    pad = (weight.shape[-1] - 1) // 2
    if pad:
        x = F.pad(x, (pad, pad), mode="circular")
    return F.conv1d(x, weight, bias, stride=stride)


def _ln(x, w, p):
    return F.layer_norm(x, (x.shape[-1],), w[p + ".weight"], w[p + ".bias"])


def residual_block(x, w, p, stride):
    skip = F.conv1d(x, w[p + ".skip.weight"], None, stride=stride)
    y = _ln(F.silu(_circ_conv(x, w[p + ".conv3x3_l1.weight"], stride=stride)), w, p + ".conv3x3_l1_norm")
    y = _ln(F.silu(_circ_conv(y, w[p + ".conv3x3_l2.weight"])), w, p + ".conv3x3_l2_norm")
    return _ln(y + skip, w, p + ".skip_norm")


def encoder(x, w, name):
    """[B, 1, N] -> [B, C, N/4]: three residual blocks, strides 2, 2, 1."""
    p = f"{name}.model"
    for i, s in enumerate((2, 2, 1)):
        x = residual_block(x, w, f"{p}.block_l{i}", s)
    return x


def decoder(z, w):
    """[B, 16, N/4] -> [B, 1, N]."""
    p = "state_decoder.model"
    for i in (0, 1):
        z = F.conv_transpose1d(z, w[f"{p}.block_l{i}.deconvolution.weight"], w[f"{p}.block_l{i}.deconvolution.bias"],
                               stride=2, padding=1, output_padding=1)
        z = _ln(F.silu(z), w, f"{p}.block_l{i}.layernorm")
    z = _ln(F.silu(_circ_conv(z, w[f"{p}.block_l2.convolution.weight"], w[f"{p}.block_l2.convolution.bias"])), w,
            f"{p}.block_l2.layernorm")
    return _circ_conv(z, w[f"{p}.block_l3.convolution.weight"], w[f"{p}.block_l3.convolution.bias"])


def lstm_cell(x, h, c, w):
    p = "transition_model.cnnlstmcell"
    g = lambda n: _circ_conv(x, w[f"{p}.Wx{n}.weight"], w[f"{p}.Wx{n}.bias"]) + _circ_conv(h, w[f"{p}.Wh{n}.weight"])
    i, f = torch.sigmoid(g("i")), torch.sigmoid(g("f"))
    c_new = f * c + i * torch.tanh(g("c"))
    o = torch.sigmoid(g("o"))
    return o * torch.tanh(c_new), c_new


def _fold(fn, x, *args):
    b, t = x.shape[:2]
    y = fn(x.reshape(b * t, *x.shape[2:]), *args)
    return y.reshape(b, t, *y.shape[1:])


def rollout(w, states, actions, hidden, delta, dscale=None):
    """One chunk: one action per step (times = k*delta, targets = (k+1)*delta).  ``dscale`` maps the
    decoded scaled delta back to physical units (Normalize.Inverse with scalar stats) or is None."""
    lstates = _fold(encoder, states, w, "state_encoder")
    lactions = _fold(encoder, actions, w, "action_encoder")
    n_given, n_steps = states.shape[1], actions.shape[1]
    if hidden is None:
        B = states.shape[0]
        hidden = (w["transition_model.H0"].repeat(B, 1, 1), w["transition_model.C0"].repeat(B, 1, 1))
    H, C = hidden
    outputs, deltas = [], []
    output = states[:, 0]
    for k in range(n_steps):
        if k < n_given:
            h_in, base = lstates[:, k], states[:, k]   # teacher forcing replaces H (transition.py:276)
        else:
            h_in, base = H, output                     # free running carries the cell's own H
            # (transition.py:285-296 ignores its `states` argument, so the re-encoded prediction
            #  `inlast` of surrogate.py:103,115 never enters the arithmetic)
        H, C = lstm_cell(lactions[:, k], h_in, C, w)
        d = decoder(H, w)
        output = base + delta * (d if dscale is None else dscale(d))
        outputs.append(output)
        deltas.append(d)
    return torch.stack(outputs, 1), torch.stack(deltas, 1), (H, C)


def tbptt_loss(w, states, actions, delta=0.25, tau=5, tbtt=10, dscale=None, undscale=None):
    """training.py:64-112 in delta mode: returns (loss, hsteploss, outputs, outdeltas)."""
    outs, dels = [], []
    seed, hidden = states[:, :tau], None
    for achunk in torch.split(actions, tbtt, dim=1):
        o, d, hidden = rollout(w, seed, achunk, hidden, delta, dscale)
        outs.append(o)
        dels.append(d)
        seed = o[:, -1:].detach()
        hidden = tuple(h.detach() for h in hidden)
    outputs = torch.cat(outs, 1)
    outdeltas = torch.cat(dels, 1)[:, :-1]
    target = torch.diff(states, dim=1) / delta
    if undscale is not None:
        target = undscale(target)
    err = (outdeltas - target) ** 2
    return err.mean(), err.mean(dim=(0, 2, 3)), outputs, outdeltas
