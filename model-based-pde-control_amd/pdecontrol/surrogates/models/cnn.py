"""1-D convolutional building blocks of the surrogate (periodic domain -> circular padding).

API / state_dict mirror of the reference's ``pdecontrol/surrogates/models/cnn.py``:
``ConvBlock`` :6-41, ``DeConvolutionBlock`` :44-70, ``ResidualBlock`` :73-145, ``ConvNet`` :148-173.
Sub-module names (``convolution``, ``deconvolution``, ``layernorm``, ``conv3x3_l1``,
``conv3x3_l1_norm``, ``conv3x3_l2``, ``conv3x3_l2_norm``, ``skip``, ``skip_norm``, ``block_l<i>``)
and their construction order are kept, so reference checkpoints load and ``torch.manual_seed``
reproduces the reference's initial weights.  On CUDA tensors the blocks run the fused HIP kernels
of ``pdecontrol.surrogates.ops`` when they are enabled; everything else is plain torch.
"""
from copy import deepcopy

from torch import nn

from pdecontrol.surrogates import ops


def _same_padding(kernel_size):
    return int((kernel_size - 1) / 2)


class ConvBlock(nn.Module):
    """Conv1d (circular) -> activation -> optional LayerNorm over the spatial axis."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, padding_mode="circular", bias=True,
                 activation=nn.ELU, layernorm=None, **kwargs):
        super().__init__()
        self.convolution = nn.Conv1d(in_channels, out_channels, kernel_size, stride, padding,
                                     padding_mode=padding_mode, bias=bias)
        self.layernorm = layernorm
        self.activation = activation()

    def forward(self, input):
        return ops.conv_act_norm(input, self.convolution, self.activation, self.layernorm)


class DeConvolutionBlock(nn.Module):
    """ConvTranspose1d (zero padded) -> activation -> optional LayerNorm."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=2, bias=True, activation=nn.ELU,
                 layernorm=None, **kwargs):
        super().__init__()
        self.deconvolution = nn.ConvTranspose1d(in_channels, out_channels, kernel_size, stride, bias=bias, **kwargs)
        self.layernorm = layernorm
        self.activation = activation()

    def forward(self, input):
        return ops.conv_act_norm(input, self.deconvolution, self.activation, self.layernorm)


class ResidualBlock(nn.Module):
    """y = LN(act(conv3(x, stride))); y = LN(act(conv3(y))); out = LN(y + conv1x1(x, stride))."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=2, padding_mode="circular", bias=False,
                 activation=nn.ELU, layernorm=None, **kwargs):
        super().__init__()
        pad = (_same_padding(kernel_size),)
        self.conv3x3_l1 = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, bias=bias,
                                    padding=pad, padding_mode=padding_mode)
        self.conv3x3_l1_norm = deepcopy(layernorm)
        self.conv3x3_l2 = nn.Conv1d(out_channels, out_channels, kernel_size=kernel_size, stride=1, bias=bias,
                                    padding=pad, padding_mode=padding_mode)
        self.conv3x3_l2_norm = deepcopy(layernorm)
        self.skip = nn.Conv1d(in_channels, out_channels, kernel_size=1, stride=stride, bias=bias, padding=(0,),
                              padding_mode=padding_mode)
        self.skip_norm = deepcopy(layernorm)
        self.activation = activation()

    def forward(self, x):
        shortcut = self.skip(x)
        y = ops.conv_act_norm(x, self.conv3x3_l1, self.activation, self.conv3x3_l1_norm)
        y = ops.conv_act_norm(y, self.conv3x3_l2, self.activation, self.conv3x3_l2_norm)
        y = y + shortcut
        return y if self.skip_norm is None else self.skip_norm(y)


class ConvNet(nn.Module):
    """Stack of blocks; per-block keyword lists are zipped by index (a list shorter than the block
    index simply does not contribute that keyword), channels chain through ``out_channels``."""

    def __init__(self, in_channels, blocks, **kwargs):
        super().__init__()
        self.layers = []
        for idx, block_cls in enumerate(blocks):
            block_kwargs = {name: values[idx] for name, values in kwargs.items() if len(values) > idx}
            name = f"block_l{idx}"
            setattr(self, name, block_cls(in_channels=in_channels, **block_kwargs))
            self.layers.append(name)
            in_channels = kwargs["out_channels"][idx]

    def forward(self, inputs):
        assert inputs.dim() == 3, "ConvNet expects [batch, channels, height]"
        for name in self.layers:
            inputs = getattr(self, name)(inputs)
        return inputs
