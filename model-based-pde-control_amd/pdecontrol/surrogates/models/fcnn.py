"""Fully connected block used by the locality-ablation factories (mirror of the reference's
``pdecontrol/surrogates/models/fcnn.py`` :5-28)."""
from torch import nn


class LinearBlock(nn.Module):
    def __init__(self, in_channels, in_size, out_channels, out_size, activation=nn.LeakyReLU):
        super().__init__()
        self.in_channels, self.in_size = in_channels, in_size
        self.out_channels, self.out_size = out_channels, out_size
        self.linear = nn.Linear(in_channels * in_size, out_channels * out_size)
        self.activation = activation()

    def forward(self, states):
        flat = states.reshape(states.size(0), self.in_channels * self.in_size)
        return self.activation(self.linear(flat)).reshape(states.size(0), self.out_channels, self.out_size)
