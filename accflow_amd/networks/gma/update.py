"""GMAUpdateBlock (reference networks/gma/update.py:112-139): RAFT's update block with the GRU input
extended by the globally aggregated motion features (512 input channels)."""
import torch
import torch.nn as nn

from ... import ops
from .._packs import PackCache, require_cuda
from ..raft.update import BasicMotionEncoder, BasicUpdateBlock, FlowHead, SepConvGRU, UpdateWorkspace  # noqa: F401
from .modules import Aggregate


class GMAUpdateBlock(BasicUpdateBlock):
    def __init__(self, args, hidden_dim=128):
        nn.Module.__init__(self)
        self.args = args
        self.encoder = BasicMotionEncoder(args)
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim + hidden_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=256)
        self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(256, 64 * 9, 1, padding=0))
        self.aggregator = Aggregate(args=self.args, dim=128, dim_head=128, heads=self.args.num_heads)
        self._packs = PackCache()

    def prepack(self):
        super().prepack()
        self.aggregator._packs.conv("v", self.aggregator.to_v)

    def aggregate(self, ws, attention):
        """motion_features_global -> the tail slice of the GRU input [inp | motion | motion_global] (update.py:131-135)"""
        hd = ws.hidden
        mg = ws.hx[:, hd + 256:hd + 384]
        if ws.s16 and getattr(attention, "attn16", None) is not None:
            self.aggregator.forward_ws(attention, ws, None, ws.x16.channels(128, 256))   # (only the GRU convs read it)
            return
        self.aggregator(attention, ws.motion.contiguous(), out=mg)
        if ws.s16:   # the GRU convolutions read the S16 form
            ops.to_s16(mg, ws.x16.channels(128, 256))

    def step(self, ws, coords1, want_mask, attention=None):
        self.motion_encoder(ws)
        self.aggregate(ws, attention)
        self.gru_step(ws)
        self.flow_delta(ws, coords1=coords1)
        return self.up_mask(ws) if want_mask else None

    @torch.no_grad()
    @ops.range_guarded
    def forward(self, net, inp, corr, flow, attention):
        require_cuda(net, inp, corr, flow, attention)
        B, _, h, w = net.shape
        ws = UpdateWorkspace(B, h, w, net.device, x_dim=384)
        ops.copy_into(net.float(), ws.net)
        ops.copy_into(inp.float(), ws.inp)
        ops.copy_into(corr.float(), ws.corr)
        ws.fill_s16_inputs(flow.float().contiguous(), is_flow=True)
        if not ws.s16:
            ops.flow_from_coords(flow.float().contiguous(), dst0=ws.flow, dst1=ws.motion_flow, stack16=ws.flow16, is_flow=True)
        self.motion_encoder(ws)
        self.aggregate(ws, attention)
        self.gru_step(ws)
        delta = self.flow_delta(ws)
        mask = self.up_mask(ws)
        return ws.net32(), mask, delta
