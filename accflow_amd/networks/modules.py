"""ZeroConv2d (reference networks/modules.py:81-97): out = conv3x3(x) * exp(3 * scale).

On the HIP path exp(3*scale) is folded into the packed weights and bias (AccPlus does the same through
`out_scale()` so that the sigmoid of the mask channels can follow in place); parameters keep the reference's
names `conv.weight`, `conv.bias`, `scale`."""
import torch
import torch.nn as nn

from .. import ops
from ._packs import PackCache, require_cuda


class ZeroConv2d(nn.Module):
    def __init__(self, in_channel, out_channel):
        super().__init__()
        self.conv = nn.Conv2d(in_channel, out_channel, 3, padding=1)
        self.conv.weight.data.zero_()
        self.conv.bias.data.zero_()
        self.scale = nn.Parameter(torch.zeros(1, out_channel, 1, 1))
        self._packs = PackCache()

    def out_scale(self):
        return torch.exp(self.scale.detach().float() * 3).reshape(-1)

    @torch.no_grad()
    def forward(self, x):
        """modules.py:94-97"""
        require_cuda(x)
        return ops.conv2d(self._packs.conv("z", self.conv, scale=self.out_scale, scale_dep=self.scale), x.float().contiguous())
