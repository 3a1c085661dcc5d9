"""Autoregressive surrogate architectures (mirror of the reference's
``pdecontrol/architectures/autoreg.py``: ``KSAutoRegFullyConnectedLSTM`` :10-41,
``KSAutoRegConvolutionalLSTM`` :44-101), plus ``KSAutoRegConvolutionalLSTMN`` -- the same network
with its LayerNorm widths derived from the grid size N, because the reference hard-codes N = 64
and BASELINE configs 3/4 run N = 256."""
from torch import nn

from pdecontrol.surrogates.factory import PDESurrogateFactory
from pdecontrol.surrogates.models import cnn as CNN
from pdecontrol.surrogates.models.fcnn import LinearBlock
from pdecontrol.surrogates.surrogate import AutoRegPDESurrogate
from pdecontrol.surrogates.transition import CNNLSTMTransitionModel, LSTMTransitionModel


class KSAutoRegFullyConnectedLSTM(PDESurrogateFactory):
    """Spatial & temporal locality ablation: dense encoder/decoder around a flat LSTM."""

    def surrogate(self, **kwargs):
        return AutoRegPDESurrogate(**kwargs)

    def model(self, **kwargs):
        state_encoder = nn.Sequential(LinearBlock(1, 64, 1, 32, activation=nn.SiLU),
                                      LinearBlock(1, 32, 1, 16, activation=nn.SiLU))
        state_decoder = nn.Sequential(LinearBlock(1, 16, 1, 32, activation=nn.SiLU),
                                      LinearBlock(1, 32, 1, 64, activation=nn.Tanh))
        action_encoder = nn.Identity()
        transition_model = LSTMTransitionModel(schannels=1, ssize=16, achannels=1, asize=4)
        return {"state_encoder": state_encoder, "state_decoder": state_decoder, "action_encoder": action_encoder,
                "transition_model": transition_model}


def _conv_lstm_model(N):
    """1->8->16->16 residual state encoder (N -> N/4), 1->2->4->4 action encoder, ConvLSTM(16 ch, N/4),
    decoder: two transposed convs (N/4 -> N) + 7-tap and 5-tap circular convs.  Construction order
    (state encoder, action encoder, transition, decoder) matches the reference so that a given
    torch seed yields the same initial weights."""
    half, quarter = N // 2, N // 4

    def encoder(channels):
        return CNN.ConvNet(
            in_channels=1, blocks=[CNN.ResidualBlock] * 3, out_channels=channels, kernel_size=[3, 3, 3],
            stride=[2, 2, 1], activation=[nn.SiLU] * 3,
            layernorm=[nn.LayerNorm(half), nn.LayerNorm(quarter), nn.LayerNorm(quarter)])

    state_encoder = encoder([8, 16, 16])
    action_encoder = encoder([2, 4, 4])
    transition_model = CNNLSTMTransitionModel(schannels=16, ssize=quarter, achannels=4, asize=quarter)
    state_decoder = CNN.ConvNet(
        in_channels=16,
        blocks=[CNN.DeConvolutionBlock, CNN.DeConvolutionBlock, CNN.ConvBlock, CNN.ConvBlock],
        out_channels=[16, 8, 1, 1], kernel_size=[3, 3, 7, 5], stride=[2, 2, 1, 1], padding=[1, 1, 3, 2],
        output_padding=[1, 1], activation=[nn.SiLU, nn.SiLU, nn.SiLU, nn.Identity],
        layernorm=[nn.LayerNorm(half), nn.LayerNorm(N), nn.LayerNorm(N)])
    return {"state_encoder": state_encoder, "state_decoder": state_decoder, "action_encoder": action_encoder,
            "transition_model": transition_model}


class KSAutoRegConvolutionalLSTM(PDESurrogateFactory):
    """The paper's model; sizes fixed for N = 64 (9 739 trainable parameters)."""

    def surrogate(self, **kwargs):
        return AutoRegPDESurrogate(**kwargs)

    def model(self, **kwargs):
        return _conv_lstm_model(64)


class KSAutoRegConvolutionalLSTMN(PDESurrogateFactory):
    """Same architecture for any grid size N divisible by 4 (taken from the scenario's ``N``)."""

    def surrogate(self, **kwargs):
        return AutoRegPDESurrogate(**kwargs)

    def model(self, N=64, **kwargs):
        assert N % 4 == 0
        return _conv_lstm_model(int(N))
