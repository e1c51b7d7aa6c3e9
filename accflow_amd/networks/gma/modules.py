"""GMA attention / aggregation on HIP kernels (reference networks/gma/modules.py).

Only the content branch exists on the AccFlow path: build_flow_estimator fixes position_only =
position_and_content = False and num_heads = 1 (networks/__init__.py:14-19).  RelPosEmb is kept as a
parameter/buffer container because its tensors are part of the reference state_dict
(`att.pos_emb.rel_ind`, `.rel_height.weight`, `.rel_width.weight`) even though they are never evaluated.
"""
import torch
import torch.nn as nn

from ... import ops
from .._packs import PackCache, require_cuda


class RelPosEmb(nn.Module):
    def __init__(self, max_pos_size, dim_head):
        super().__init__()
        self.rel_height = nn.Embedding(2 * max_pos_size - 1, dim_head)
        self.rel_width = nn.Embedding(2 * max_pos_size - 1, dim_head)
        idx = torch.arange(max_pos_size)
        self.register_buffer("rel_ind", idx.view(1, -1) - idx.view(-1, 1) + max_pos_size - 1)


class Attention(nn.Module):
    def __init__(self, *, args, dim, max_pos_size=100, heads=4, dim_head=128):
        super().__init__()
        self.args = args
        self.heads = heads
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        self.to_qk = nn.Conv2d(dim, heads * dim_head * 2, 1, bias=False)
        self.pos_emb = RelPosEmb(max_pos_size, dim_head)
        self._packs = PackCache()

    @torch.no_grad()
    @ops.range_guarded
    def forward(self, fmap):
        """(B, dim, h, w) -> attn (B, heads=1, h*w, h*w) = softmax_j(scale * <q_i, k_j>)  (modules.py:54-76)"""
        if self.heads != 1 or self.args.position_only or self.args.position_and_content:
            raise NotImplementedError("only the single-head content attention is on the AccFlow path")
        require_cuda(fmap)
        qk = ops.conv2d(self._packs.conv("qk", self.to_qk), fmap.float())
        return ops.gma_attention(qk, self.dim_head, self.scale)

    @torch.no_grad()
    def forward_t(self, fmap, runs=None):
        """Same attention, transposed storage, for Aggregate's matrix-core path (internal to RAFTGMA).  In S16 mode the
        matrix is stored pre-split (ops.gma_attention_s16): it is the aggregation GEMM's activation operand in all 12
        refinement iterations and is then never converted again."""
        require_cuda(fmap)
        qk = ops.conv2d(self._packs.conv("qk", self.to_qk), fmap.float())
        if ops.s16_active() and self.dim_head % 32 == 0:
            return TransposedAttention(None, runs, attn16=ops.gma_attention_s16(qk, self.dim_head, self.scale))
        return TransposedAttention(ops.gma_attention_t(qk, self.dim_head, self.scale), runs)


class TransposedAttention:
    """Hot-path handle: the attention stored j-major (see ops.gma_attention_t)."""

    def __init__(self, attn_t, runs=None, attn16=None):
        self.attn_t = attn_t
        self.attn16 = attn16   # ops.S16 (G, P, h, w): the pre-split form (then attn_t is None)
        self.runs = runs  # None: attn_t[b] belongs to item b; else [(g, b0, b1)]: items b0..b1-1 all use attn_t[g]


class Aggregate(nn.Module):
    def __init__(self, args, dim, heads=4, dim_head=128):
        super().__init__()
        self.args = args
        self.heads = heads
        self.scale = dim_head ** -0.5
        inner = heads * dim_head
        self.to_v = nn.Conv2d(dim, inner, 1, bias=False)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.project = nn.Conv2d(inner, dim, 1, bias=False) if dim != inner else None
        self._packs = PackCache()

    @torch.no_grad()
    @ops.range_guarded
    def forward(self, attn, fmap, out=None):
        """out = fmap + gamma * (attn @ to_v(fmap))   (modules.py:102-115; project is None for dim == inner)"""
        if self.heads != 1 or self.project is not None:
            raise NotImplementedError("only heads=1, dim == inner_dim is on the AccFlow path")
        fm = fmap.float().contiguous()
        require_cuda(fm)
        v = ops.conv2d(self._packs.conv("v", self.to_v), fm)
        if isinstance(attn, TransposedAttention) and attn.attn16 is not None:
            if out is None:
                out = torch.empty_like(fm)
            self._aggregate16(attn, v, fm, fm.stride(0), out, None)
            return out
        if isinstance(attn, TransposedAttention):
            if attn.runs is None:
                return ops.gma_aggregate_t(attn.attn_t, v, fm, self.gamma, out=out)
            # items that share one attention matrix are stacked along the output rows of ONE GEMM (the shared
            # matrix is the GEMM's activation operand: read once for the whole run)
            if out is None:
                out = torch.empty_like(fm)
            _, D, h, w = fm.shape
            for g, b0, b1 in attn.runs:
                n = b1 - b0
                if n == 1:
                    ops.gma_aggregate_t(attn.attn_t[g:g + 1], v[b0:b1], fm[b0:b1], self.gamma, out=out[b0:b1])
                else:
                    rows = ops.gma_aggregate_t(attn.attn_t[g:g + 1], v[b0:b1].view(1, n * D, h, w),
                                               fm[b0:b1].view(1, n * D, h, w), self.gamma)
                    out[b0:b1].copy_(rows.view(n, D, h, w))
            return out
        require_cuda(attn)
        return ops.gma_aggregate(attn, v, fm, self.gamma, out=out)

    def _aggregate16(self, attn, v, fmap, fmap_bs, out, out16):
        """One GEMM per run of items sharing an attention matrix; residual read from / results written into the items'
        slices in place (fmap / out: (B, D, h, w) views with any batch stride, out16: ops.S16 or None)."""
        B, D, h, w = v.shape
        runs = attn.runs if attn.runs is not None else [(b, b, b + 1) for b in range(B)]
        a16 = attn.attn16
        item = a16.bs * 4
        for g, b0, b1 in runs:
            ops.gma_aggregate_s16(a16.ptr() + g * item, v[b0:b1], fmap[b0].data_ptr(), fmap_bs,
                                  self.gamma, out[b0].data_ptr() if out is not None else None, out.stride(0) if out is not None else 0,
                                  out16.batch(b0, b1).ptr() if out16 is not None else None, out16.bs if out16 is not None else 0,
                                  b1 - b0, D, h, w)

    @torch.no_grad()
    def forward_ws(self, attn, ws, mg32, mg16):
        """Hot path on the update workspace (S16 mode): v from the pre-split motion features, the aggregated features
        straight into the GRU input's fp32 slice mg32 and its S16 form mg16 - no copies, no conversion pass."""
        v = ops.conv2d(self._packs.conv("v", self.to_v), ws.motion16)
        self._aggregate16(attn, v, ws.motion, ws.hx.stride(0), mg32, mg16)
