"""BasicEncoder / ResidualBlock on HIP kernels.

Parameter layout (names, shapes, the norm3 <-> downsample[1] sharing) follows the reference's
networks/raft/extractor.py:5-63 (ResidualBlock) and :115-225 (BasicEncoder) so its checkpoints load
with strict=True; gma/extractor.py:115-188 is the same network.  forward() never calls a torch conv:
convs go to accflow_conv2d_f32 (BatchNorm-eval folded into the packed weights, ReLU / residual add in
the conv epilogue) and InstanceNorm to accflow_instance_norm_f32.
"""
import torch
import torch.nn as nn

from ... import ops
from .._packs import PackCache, require_cuda, span

import os

USE_S16_ENCODER = os.environ.get("ACCFLOW_S16_ENCODER", "1") == "1"   # (0: the round-3 encoder path, A/B)
USE_STEM_KERNEL = os.environ.get("ACCFLOW_CONV_STEM", "1") == "1"     # (0: stem on the im2col kernel + a to_s16 pass, A/B)
# the 1x1 stride-2 projection of a block's input rides in the block's strided 3x3 launch (accflow_conv_desc.split_c0); 0: a
# launch of its own (round 4-5, A/B)
FUSE_PROJECTION = os.environ.get("ACCFLOW_FUSE_PROJECTION", "1") == "1"

_NORMS = {
    "group": lambda ch, groups: nn.GroupNorm(num_groups=groups, num_channels=ch),
    "batch": lambda ch, groups: nn.BatchNorm2d(ch),
    "instance": lambda ch, groups: nn.InstanceNorm2d(ch),
    "none": lambda ch, groups: nn.Sequential(),
}


def _make_norm(kind, ch, groups):
    if kind not in _NORMS:
        raise ValueError("unknown norm_fn %r" % (kind,))
    return _NORMS[kind](ch, groups)


class ResidualBlock(nn.Module):
    def __init__(self, in_planes, planes, norm_fn="group", stride=1):
        super().__init__()
        self.norm_fn = norm_fn
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        projected = stride != 1 or in_planes != planes
        for name in ("norm1", "norm2") + (("norm3",) if projected else ()):
            setattr(self, name, _make_norm(norm_fn, planes, planes // 8))
        self.downsample = None
        if projected:
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride), self.norm3)

    def _out_hw(self, H, W):
        st = self.conv1.stride[0]
        return ((H + 2 - 3) // st + 1, (W + 2 - 3) // st + 1)

    def run16(self, x16, packs, tag):
        """The block on PRE-SPLIT activations (ops.S16, f16x3 mode; multi-source kernel csrc/conv_s16m_kernel.h): input and
        output exist as S16 tensors ONLY - the convolutions stage them by LDS DMA, the residual add reads (hi + lo) / 2^4
        (accflow_conv_desc.e0_fmt / the S16-residual InstanceNorm pass): the same 4 bytes per element as fp32, 22 bits.
        Stride-2 convolutions (conv1 and the 1x1 downsample of the first block of layer2 / layer3, extractor.py:9,52) run
        as stride-1 work over the parity classes of x16."""
        kind = self.norm_fn
        B, _, H, W = x16.shape
        strided = self.conv1.stride[0] == 2
        OH, OW = self._out_hw(H, W)
        dev = x16.device
        planes = self.conv1.out_channels
        out16 = ops.S16.empty(B, planes, OH, OW, dev)
        fuse_proj = (FUSE_PROJECTION and strided and self.downsample is not None and planes in (64, 96, 128)
                     and tuple(self.downsample[0].kernel_size) == (1, 1) and self.downsample[0].stride[0] == 2)
        if kind == "instance" and fuse_proj and ops.USE_NORM_ON_LOAD and ops.USE_NORM_STATS:
            # conv1 (3x3, stride 2) and the projection of the block input (1x1, stride 2, extractor.py:52-53) in ONE launch
            # (accflow_conv_desc.split_c0): raw outputs + InstanceNorm statistics of both; conv2 normalises conv1's half on
            # load, the closing pass normalises the projection's half itself (no pass of its own over it)
            pkp = packs.multi_proj(tag + ".c1p", self.conv1, self.downsample[0])
            both, stb = ops.conv2d_multi(pkp, [x16] * len(pkp.C), want_stats=True, out_hw=(OH, OW))
            r = None
            if stb is not None:
                r = ops.conv2d(packs.conv(tag + ".c2", self.conv2), both[:, :planes], want_stats=True,
                               in_norm=ops.instance_stats_finalize(stb, self.norm1.eps, c0=0, C=planes))
            if r is not None and r[1] is not None:
                return ops.instance_norm_proj(r[0], r[1], both[:, planes:], stb, planes, out16, eps=self.norm2.eps)
            # (a kernel route without statistics / normalise-on-load: the separate launches below)
        if kind == "instance":
            pk1 = packs.multi(tag + ".c1m", self.conv1, strided=strided)
            y, st = ops.conv2d_multi(pk1, [x16] * len(pk1.C), want_stats=True, out_hw=(OH, OW))
            r = None
            if st is not None and ops.USE_NORM_ON_LOAD:
                r = ops.conv2d(packs.conv(tag + ".c2", self.conv2), y, want_stats=True,
                               in_norm=ops.instance_stats_finalize(st, self.norm1.eps))
            if r is None:
                ops.instance_norm(y, 1, eps=self.norm1.eps, stats=st)
                r = ops.conv2d(packs.conv(tag + ".c2", self.conv2), y, want_stats=True)
            y2, st2 = r
            res = x16
            if self.downsample is not None:
                pkd = packs.multi(tag + ".dsm", self.downsample[0], strided=strided)
                xd, st3 = ops.conv2d_multi(pkd, [x16] * len(pkd.C), want_stats=True, out_hw=(OH, OW))
                res = ops.instance_norm(xd, 0, eps=self.norm3.eps, stats=st3)          # (fp32, in place)
            ops.instance_norm(y2, 2, res=res, eps=self.norm2.eps, stats=st2, out16=out16, fp32_out=False)
            return out16
        bn = kind == "batch"
        if fuse_proj:
            # conv1 (3x3, stride 2) and the projection of the block input (1x1, stride 2, extractor.py:52-53) in ONE launch:
            # the 3x3's parity class (0, 0) reads exactly the pixels the projection reads; channels [0, planes) =
            # relu(conv1), [planes, 2 planes) = the projection (no activation)
            pkp = packs.multi_proj(tag + ".c1p", self.conv1, self.downsample[0], bn=self.norm1 if bn else None,
                                   bn_proj=self.norm3 if bn else None)
            both = ops.S16.empty(B, 2 * planes, OH, OW, dev)
            ops.conv2d_multi(pkp, [x16] * len(pkp.C), act=ops.ACT_RELU, out16=both, fp32_out=False, out_hw=(OH, OW))
            pk2 = packs.multi(tag + ".c2m", self.conv2, bn=self.norm2 if bn else None)
            ops.conv2d_multi(pk2, [both.channels(0, planes)], act=ops.ACT_RELU, epi=ops.EPI_RES_RELU,
                             e0=both.channels(planes, 2 * planes), out16=out16, fp32_out=False)
            return out16
        pk1 = packs.multi(tag + ".c1m", self.conv1, bn=self.norm1 if bn else None, strided=strided)
        y16 = ops.S16.empty(B, planes, OH, OW, dev)
        ops.conv2d_multi(pk1, [x16] * len(pk1.C), act=ops.ACT_RELU, out16=y16, fp32_out=False, out_hw=(OH, OW))
        res = x16
        if self.downsample is not None:
            pkd = packs.multi(tag + ".dsm", self.downsample[0], bn=self.norm3 if bn else None, strided=strided)
            res = ops.S16.empty(B, planes, OH, OW, dev)
            ops.conv2d_multi(pkd, [x16] * len(pkd.C), out16=res, fp32_out=False, out_hw=(OH, OW))
        pk2 = packs.multi(tag + ".c2m", self.conv2, bn=self.norm2 if bn else None)
        ops.conv2d_multi(pk2, [y16], act=ops.ACT_RELU, epi=ops.EPI_RES_RELU, e0=res, out16=out16, fp32_out=False)
        return out16

    def run(self, x, packs, tag):
        kind = self.norm_fn
        if kind == "group":
            raise NotImplementedError("GroupNorm encoders are not on the AccFlow inference path")
        bn = kind == "batch"
        if kind == "instance":
            # (the convolutions gather the per-plane statistics in their epilogues: each norm is one pass, not three)
            y, st = ops.conv2d(packs.conv(tag + ".c1", self.conv1), x, want_stats=True)
            r = None
            if st is not None and ops.USE_NORM_ON_LOAD:
                # relu(norm1(y)) is consumed by conv2 only: conv2's patch loader normalises on the way into LDS
                r = ops.conv2d(packs.conv(tag + ".c2", self.conv2), y, want_stats=True,
                               in_norm=ops.instance_stats_finalize(st, self.norm1.eps))
            if r is None:
                ops.instance_norm(y, 1, eps=self.norm1.eps, stats=st)
                r = ops.conv2d(packs.conv(tag + ".c2", self.conv2), y, want_stats=True)
            y2, st2 = r
            if self.downsample is not None:
                x, st3 = ops.conv2d(packs.conv(tag + ".ds", self.downsample[0]), x, want_stats=True)
                ops.instance_norm(x, 0, eps=self.norm3.eps, stats=st3)
            return ops.instance_norm(y2, 2, res=x, eps=self.norm2.eps, stats=st2)
        pk1 = packs.conv(tag + ".c1", self.conv1, bn=self.norm1 if bn else None)
        if ops.s16_active() and pk1.stride == 1 and pk1.wpatch16 is not None:
            # relu(conv1(x)) is read by conv2 only: it goes out PRE-SPLIT (ops.S16, no fp32 copy - the same bytes) and
            # conv2 stages it by LDS DMA instead of gathering and splitting it per tile
            B, _, H, W = x.shape
            y = ops.S16.empty(B, pk1.Cout, H, W, x.device)
            ops.conv2d(pk1, x, act=ops.ACT_RELU, out16=y, fp32_out=False)
        else:
            y = ops.conv2d(pk1, x, act=ops.ACT_RELU)
        if self.downsample is not None:
            x = ops.conv2d(packs.conv(tag + ".ds", self.downsample[0], bn=self.norm3 if bn else None), x)
        return ops.conv2d(packs.conv(tag + ".c2", self.conv2, bn=self.norm2 if bn else None), y,
                          act=ops.ACT_RELU, epi=ops.EPI_RES_RELU, e0=x)


class BasicEncoder(nn.Module):
    def __init__(self, input_dim=3, output_dim=128, norm_fn="batch", dropout=0.0):
        super().__init__()
        self.norm_fn = norm_fn
        self.norm1 = _make_norm(norm_fn, 64, 8)
        self.conv1 = nn.Conv2d(input_dim, 64, kernel_size=7, stride=2, padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        widths, strides, prev = (64, 96, 128), (1, 2, 2), 64
        for idx, (wd, st) in enumerate(zip(widths, strides), start=1):
            setattr(self, "layer%d" % idx, nn.Sequential(ResidualBlock(prev, wd, norm_fn, stride=st),
                                                         ResidualBlock(wd, wd, norm_fn, stride=1)))
            prev = wd
        self.in_planes = prev
        self.conv2 = nn.Conv2d(prev, output_dim, kernel_size=1)
        self.dropout = nn.Dropout2d(p=dropout) if dropout > 0 else None
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.GroupNorm)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        self._packs = PackCache()

    def _check_mode(self):
        if self.training and (self.norm_fn == "batch" or self.dropout is not None):
            raise RuntimeError("BasicEncoder: the HIP path is inference-only (call .eval()); batch-stat "
                               "BatchNorm / dropout are training features outside the AccFlow inference path")

    @torch.no_grad()
    @ops.range_guarded
    def forward(self, x, want16=False):
        """want16 (f16x3 mode; AccFlow's context encoder, whose output both getOcc - fp32 - and AccPlus's convolutions -
        pre-split - read): returns (fp32 outputs, ops.S16 outputs) - the second None when the S16 path is off."""
        is_list = isinstance(x, (tuple, list))
        if is_list:
            batch_dim = x[0].shape[0]
            x = span(x)      # (no copy when the frames are consecutive views of one stacked tensor: AccFlow.forward)
        require_cuda(x)
        self._check_mode()
        x = x.float().contiguous()
        pk = self._packs
        s16 = ops.s16_active() and USE_S16_ENCODER and self.norm_fn != "group"
        x16 = None
        if self.norm_fn == "instance":
            x, st = ops.conv2d(pk.conv("stem", self.conv1), x, want_stats=True)
            if s16:
                x16 = ops.S16.empty(x.shape[0], x.shape[1], x.shape[2], x.shape[3], x.device)
            ops.instance_norm(x, 1, eps=self.norm1.eps, stats=st, out16=x16, fp32_out=not s16)
        else:
            pks = pk.conv("stem", self.conv1, bn=self.norm1 if self.norm_fn == "batch" else None)
            if s16 and USE_STEM_KERNEL and pks.wsplit16 is not None and tuple(self.conv1.weight.shape[1:]) == (3, 7, 7):
                # the stem kernel writes relu(conv) pre-split only (csrc/conv_stem.hip): no fp32 round trip, no to_s16 pass
                OH, OW = pks.out_size(x.shape[2], x.shape[3])
                x16 = ops.S16.empty(x.shape[0], pks.Cout, OH, OW, x.device)
                ops.conv2d(pks, x, act=ops.ACT_RELU, out16=x16, fp32_out=False)
            else:
                x = ops.conv2d(pks, x, act=ops.ACT_RELU)
                if s16:
                    x16 = ops.to_s16(x)
        if s16:
            for li in (1, 2, 3):
                for bi, blk in enumerate(getattr(self, "layer%d" % li)):
                    x16 = blk.run16(x16, pk, "l%d.%d" % (li, bi))
            o16 = ops.S16.empty(x16.shape[0], self.conv2.out_channels, x16.shape[2], x16.shape[3], x16.device) if want16 else None
            x = ops.conv2d_multi(pk.multi("headm", self.conv2), [x16], out16=o16)
        else:
            o16 = None
            for li in (1, 2, 3):
                for bi, blk in enumerate(getattr(self, "layer%d" % li)):
                    x = blk.run(x, pk, "l%d.%d" % (li, bi))
            x = ops.conv2d(pk.conv("head", self.conv2), x)
        if is_list:
            x = torch.split(x, batch_dim, dim=0)
            if want16 and o16 is not None:
                o16 = [o16.batch(k * batch_dim, (k + 1) * batch_dim) for k in range(len(x))]
        return (x, o16) if want16 else x
