"""RAFT + Global Motion Aggregation (reference networks/gma/gma.py:14-125) on gfx950 kernels.
Same forward signature and state_dict; execution notes as in raft/raft.py."""
import os

import torch.nn as nn

from ... import ops
from ..raft.extractor import BasicEncoder
from ..raft.raft import RAFT
from .modules import Attention
from .update import GMAUpdateBlock

SHARE_ATTENTION = os.environ.get("ACCFLOW_GMA_SHARE_ATTENTION", "1") != "0"


class RAFTGMA(RAFT):
    KEEP_CONTEXT_RUNS = True      # (the aggregation runs one GEMM per run of items sharing an attention matrix)

    def __init__(self, args):
        nn.Module.__init__(self)
        self.args = args
        self.hidden_dim = hdim = 128
        self.context_dim = cdim = 128
        args.corr_levels = 4
        args.corr_radius = 4
        if "dropout" not in self.args:
            self.args.dropout = 0
        self.fnet = BasicEncoder(output_dim=256, norm_fn="instance", dropout=args.dropout)
        self.cnet = BasicEncoder(output_dim=hdim + cdim, norm_fn="batch", dropout=args.dropout)
        self.update_block = GMAUpdateBlock(self.args, hidden_dim=hdim)
        self.att = Attention(args=self.args, dim=cdim, heads=self.args.num_heads, max_pos_size=160, dim_head=cdim)

    def _x_dim(self):
        return 384

    def _prepack(self):
        super()._prepack()
        self.att._packs.conv("qk", self.att.to_qk)

    def _prepare_context(self, ws, cnet_feat, ids=None):
        super()._prepare_context(ws, cnet_feat)
        # attention = self.att(inp), once per image1 (gma.py:96); kept on the workspace
        fast = ops.current_mode() != ops.CONV_F32  # aggregation on the split-bf16 matrix cores needs the j-major attention
        inp = ws.inp.contiguous()
        if not fast:
            ws.attention = self.att(inp)
            return
        # pairs out of the same image1 have the same attention (829 MB per item at 720x1280): build it once per
        # distinct frame, and let each run of adjacent items that share it go through ONE aggregation GEMM
        runs = None
        if ids is not None and SHARE_ATTENTION and len(set(ids)) < len(ids):
            slot, reps, runs = {}, [], []
            for b, key in enumerate(ids):
                if key not in slot:
                    slot[key] = len(reps)
                    reps.append(b)
                if runs and runs[-1][0] == slot[key]:
                    runs[-1][2] = b + 1
                else:
                    runs.append([slot[key], b, b + 1])
            inp = inp[reps].contiguous()
        ws.attention = self.att.forward_t(inp, runs=runs)

    def _iteration(self, ws, corr_fn, coords1, last):
        self._lookup_and_flow(ws, corr_fn, coords1)
        return self.update_block.step(ws, coords1, want_mask=last, attention=ws.attention)
