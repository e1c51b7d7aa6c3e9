"""ZeroConv2d parameter container (reference networks/modules.py:81-97).

out = conv3x3(x) * exp(3 * scale); on the HIP path exp(3*scale) is folded into the packed weights and
bias (see AccPlus), so this module only owns the parameters `conv.weight`, `conv.bias`, `scale`."""
import torch
import torch.nn as nn


class ZeroConv2d(nn.Module):
    def __init__(self, in_channel, out_channel):
        super().__init__()
        self.conv = nn.Conv2d(in_channel, out_channel, 3, padding=1)
        self.conv.weight.data.zero_()
        self.conv.bias.data.zero_()
        self.scale = nn.Parameter(torch.zeros(1, out_channel, 1, 1))

    def out_scale(self):
        return torch.exp(self.scale.detach().float() * 3).reshape(-1)
