"""MLP trunk (reference: lib/network/mlp.py:4-39): Linear + activation after EVERY layer, torch's
default Linear weight init (the reference's 'init' is a no-op, quirk Q11), zero biases.
Module/parameter names (`layers.<i>.weight`) match the reference's state dict."""
import torch
import torch.nn as nn

from airgym_amd.lib.network.splitk_linear import linear, linear_elu

ACTIVATIONS = {
    "tanh": torch.tanh,
    "relu": torch.relu,
    "sigmoid": torch.sigmoid,
    "elu": torch.nn.functional.elu,
    "sin": torch.sin,
}


class MLP(nn.Module):
    def __init__(self, input_size, units, activation):
        super().__init__()
        if activation not in ACTIVATIONS:
            raise ValueError(f"Unsupported activation: {activation}")
        self.activation = ACTIVATIONS[activation]
        self.activation_name = activation
        self.layers = nn.ModuleList()
        in_dim = int(input_size)
        for out_dim in units:
            layer = nn.Linear(in_dim, out_dim)
            nn.init.zeros_(layer.bias)
            self.layers.append(layer)
            in_dim = out_dim

    def forward(self, x):
        for layer in self.layers:
            if self.activation_name == "elu":
                x = linear_elu(x, layer.weight, layer.bias)
            else:
                x = self.activation(linear(x, layer.weight, layer.bias))
        return x
