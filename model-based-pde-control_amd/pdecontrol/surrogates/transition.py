"""Recurrent transition models in latent space.

API / state_dict mirror of the reference's ``pdecontrol/surrogates/transition.py``:
``TransitionModel`` :7-31, ``CNNLSTMCell`` :112-226, ``CNNLSTMTransitionModel`` :229-296 (the hot
one), ``LSTMTransitionModel`` :34-109 (fully connected ablation).  Parameter names (``Wxi`` ...
``Who``, ``H0``, ``C0``, ``cnnlstmcell``, ``lstm``) and construction order are the reference's.
The cell arithmetic itself lives in ``pdecontrol.surrogates.ops.lstm_cell`` (torch on CPU, one
fused HIP kernel for all 8 circular convolutions + gates on the GPU).
"""
import torch
from torch import nn

from pdecontrol.surrogates import ops


class TransitionModel(nn.Module):
    def __init__(self, schannels: int, ssize: int, achannels: int, asize: int, dtype=torch.FloatTensor, **kwargs):
        super().__init__()
        self.dtype = dtype
        self.schannels, self.ssize, self.achannels, self.asize = schannels, ssize, achannels, asize

    def teacherforcing(self, states, actions, hidden=None, **kwargs):
        raise NotImplementedError

    def transition(self, states, actions, hidden=None, **kwargs):
        raise NotImplementedError


class CNNLSTMCell(nn.Module):
    """i = s(Wxi*x + Whi*h), f = s(Wxf*x + Whf*h), c' = f c + i tanh(Wxc*x + Whc*h),
    o = s(Wxo*x + Who*h), h' = o tanh(c'); all convolutions k=3, circular.  The x-convolutions
    carry the biases (i, f, c: zero-initialised; o: one), the h-convolutions have none."""

    GATES = ("i", "f", "c", "o")

    def __init__(self, in_channels, out_channels, height, kernel_size=3, stride=1, bias=True,
                 padding_mode="circular"):
        super().__init__()
        self.in_channels, self.out_channels, self.height = in_channels, out_channels, height
        self.kernel_size, self.stride, self.bias = kernel_size, stride, bias
        self.padding = int((kernel_size - 1) / 2)
        for gate in self.GATES:  # creation order Wxi, Whi, Wxf, Whf, Wxc, Whc, Wxo, Who
            x_bias = bias if gate == "i" else True
            h_pad = 1 if gate == "f" else self.padding  # (sic) reference hard-codes 1 for Whf
            setattr(self, f"Wx{gate}", nn.Conv1d(in_channels, out_channels, kernel_size, stride, self.padding,
                                                 bias=x_bias, padding_mode=padding_mode))
            setattr(self, f"Wh{gate}", nn.Conv1d(out_channels, out_channels, kernel_size, 1, padding=h_pad,
                                                 bias=False, padding_mode=padding_mode))
        nn.init.zeros_(self.Wxi.bias)
        nn.init.zeros_(self.Wxf.bias)
        nn.init.zeros_(self.Wxc.bias)
        self.Wxo.bias.data.fill_(1.0)

    def forward(self, x, h, c):
        return ops.lstm_cell(x, h, c, self)


class CNNLSTMTransitionModel(TransitionModel):
    def __init__(self, schannels, ssize, achannels, asize, kernel_size=3, stride=1, bias=True, Cell=CNNLSTMCell,
                 dtype=torch.FloatTensor):
        super().__init__(schannels, ssize, achannels, asize, dtype)
        self.cnnlstmcell = Cell(in_channels=achannels, out_channels=schannels, height=ssize,
                                kernel_size=kernel_size, stride=stride, bias=bias)
        # zero, non-trainable initial hidden / cell state (kept as parameters for state_dict parity)
        self.H0 = nn.Parameter(torch.zeros(schannels, ssize).type(self.dtype), requires_grad=False)
        self.C0 = nn.Parameter(torch.zeros(schannels, ssize).type(self.dtype), requires_grad=False)

    def _initial(self, bsize):
        return self.H0.repeat(bsize, 1, 1), self.C0.repeat(bsize, 1, 1)

    def teacherforcing(self, states, actions, hidden=None, **kwargs):
        """Warm-up: H is overwritten by the encoded true state before every cell update."""
        assert states.size(1) == actions.size(1)
        H, C = self._initial(states.size(0)) if hidden is None else hidden
        outputs = []
        for t in range(states.size(1)):
            H, C = self.cnnlstmcell(actions[:, t], states[:, t], C)
            outputs.append(H)
        return torch.stack(outputs, dim=1), (H, C)

    def transition(self, states, actions, hidden, **kwargs):
        """Free running: the cell's own H is carried; ``states`` is not used."""
        H, C = hidden
        outputs = []
        for t in range(actions.size(1)):
            H, C = self.cnnlstmcell(actions[:, t], H, C)
            outputs.append(H)
        return torch.stack(outputs, dim=1), (H, C)


class LSTMTransitionModel(TransitionModel):
    """Flattened nn.LSTM over (channels x size) vectors; teacher forcing replaces H each step."""

    def __init__(self, schannels: int, ssize: int, achannels: int, asize: int, dtype=torch.FloatTensor):
        super().__init__(schannels, ssize, achannels, asize, dtype)
        width = schannels * ssize
        self.lstm = nn.LSTM(achannels * asize, width, batch_first=True)
        self.H0 = nn.Parameter(torch.zeros(width).type(self.dtype), requires_grad=False)
        self.C0 = nn.Parameter(torch.zeros(width).type(self.dtype), requires_grad=False)

    def _initial(self, bsize):
        return self.H0.repeat(bsize, 1).unsqueeze(0), self.C0.repeat(bsize, 1).unsqueeze(0)

    def teacherforcing(self, states, actions, hidden=None, **kwargs):
        bsize, steps = states.shape[:2]
        H, C = self._initial(bsize) if hidden is None else hidden
        # (.contiguous(): MIOpen's RNN refuses a strided h0, which the time-major view of [B, T, width] is; values unchanged)
        flat_s = states.reshape(bsize, steps, -1).swapaxes(0, 1).contiguous()
        flat_a = actions.reshape(bsize, steps, -1)
        outputs = []
        for t in range(steps):
            out, (H, C) = self.lstm(flat_a[:, t, None, :], (flat_s[t, None], C))
            outputs.append(out)
        outputs = torch.cat(outputs, dim=1).reshape(bsize, steps, self.schannels, self.ssize)
        return outputs, (H, C)

    def transition(self, states, actions, hidden=None, **kwargs):
        bsize, steps = actions.shape[:2]
        hidden = self._initial(bsize) if hidden is None else hidden
        outputs, hidden = self.lstm(actions.reshape(bsize, steps, -1), hidden)
        return outputs.reshape(bsize, steps, self.schannels, self.ssize), hidden
