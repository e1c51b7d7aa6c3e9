"""AccFlow backward accumulation on gfx950 kernels.

Mirrors the reference's networks/AccFlow_.py: `AccFlow(ofe).forward(images, test_mode=False)` returns
[F(2->0), F(3->0), ..., F(n-1->0)] (AccFlow_.py:157-175), `iter(I1, I2, In, F2n)` one accumulation step
(:177-201); sub-modules FlowDecoder (:13-45), FlowEncoder (:48-65), AccPlus (:68-109), Blending
(:112-124), functions getOcc (:127-135), downflow8 (:138-142); identical state_dict.

The deformable convolution (torchvision.ops.DeformConv2d in the reference, AccFlow_.py:83,104) is the
deformable mode of accflow_conv2d_f32; its parameters live in a plain container with the same names
(`dconv.weight`, `dconv.bias`).

Scheduling (exact): `forward` evaluates all estimator pairs of the sequence in ONE batched
call - they are independent because the reference never passes flow_init (AccFlow_.py:184,188) - with each
frame encoded once, and runs the context encoder once per frame; only the fusion chain is sequential.
"""
import math

import torch
import torch.nn as nn

from .. import ops
from ._packs import PackCache, require_cuda
from .modules import ZeroConv2d
from .raft.extractor import BasicEncoder


class FlowDecoder(nn.Module):
    def __init__(self, cin=128):
        super().__init__()
        self.flow = nn.Sequential(nn.Conv2d(cin, cin * 2, 3, 1, 1), nn.ReLU(True), nn.Conv2d(cin * 2, 2, 3, 1, 1))
        self.mask = nn.Sequential(nn.Conv2d(cin, cin * 2, 3, 1, 1), nn.ReLU(True), nn.Conv2d(cin * 2, 64 * 9, 1, 1, 0))
        self._packs = PackCache()

    def upsample_flow(self, flow, mask):
        require_cuda(flow, mask)
        return ops.convex_upsample(flow.float(), mask.float())

    @torch.no_grad()
    @ops.range_guarded
    def forward(self, x):
        require_cuda(x)
        pk = self._packs
        t = ops.conv2d(pk.conv("f0", self.flow[0]), x, act=ops.ACT_RELU)
        flow_small = ops.conv2d(pk.conv("f2", self.flow[2]), t)
        t = ops.conv2d(pk.conv("m0", self.mask[0]), x, act=ops.ACT_RELU, out=t)
        mask = ops.conv2d(pk.conv("m2", self.mask[2]), t)  # no 0.25 factor here (AccFlow_.py:42-43)
        return flow_small, ops.convex_upsample(flow_small, mask)


class FlowEncoder(nn.Module):
    def __init__(self, c=128):
        super().__init__()
        self.conv1 = nn.Conv2d(2, c, 7, stride=1, padding=3)
        self.conv2 = nn.Conv2d(c, c * 2, 3, stride=1, padding=1)
        self.conv3 = nn.Conv2d(c * 2, c, 1, stride=1, padding=0)
        self.relu = nn.ReLU(True)
        self._packs = PackCache()

    @torch.no_grad()
    @ops.range_guarded
    def forward(self, x):
        is_list = isinstance(x, (tuple, list))
        if is_list:
            batch_dim = x[0].shape[0]
            x = torch.cat(x, dim=0)
        require_cuda(x)
        pk = self._packs
        x = ops.conv2d(pk.conv("1", self.conv1), x.float().contiguous(), act=ops.ACT_RELU)
        x = ops.conv2d(pk.conv("2", self.conv2), x, act=ops.ACT_RELU)
        x = ops.conv2d(pk.conv("3", self.conv3), x)
        if is_list:
            x = torch.split(x, batch_dim, dim=0)
        return x


class DeformConv2d(nn.Module):
    """Parameter container with torchvision.ops.DeformConv2d's names / shapes / default init
    (3x3, stride 1, pad 1, one offset group); evaluated by the deformable mode of the HIP conv."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1):
        super().__init__()
        if (kernel_size, stride, padding) != (3, 1, 1):
            raise NotImplementedError("AccPlus uses DeformConv2d(c, c, 3, 1, 1) (AccFlow_.py:83)")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = (3, 3), (1, 1), (1, 1)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 3, 3))
        self.bias = nn.Parameter(torch.empty(out_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_channels * 9)
        nn.init.uniform_(self.bias, -bound, bound)


class AccPlus(nn.Module):
    def __init__(self, c=128):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(c * 2 + 1, c * 2, 3, 1, 1), nn.ReLU(True), nn.Conv2d(c * 2, c, 3, 1, 1))
        self.conv2 = nn.Sequential(nn.Conv2d(c * 2, c * 2, 3, 1, 1), nn.ReLU(True), nn.Conv2d(c * 2, c, 3, 1, 1),
                                   nn.ReLU(True), ZeroConv2d(c, 3 ** 3))
        self.dconv = DeformConv2d(c, c, 3, 1, 1)
        self.conv3 = nn.Sequential(nn.Conv2d(c * 2 + 1, c * 2, 3, 1, 1), nn.ReLU(True), nn.Conv2d(c * 2, c, 3, 1, 1))
        self.conv4 = nn.Sequential(nn.Conv2d(c * 4, c * 2, 3, 1, 1), nn.ReLU(True), nn.Conv2d(c * 2, c, 3, 1, 1),
                                   nn.ReLU(True), nn.Conv2d(c, c, 1, 1, 0))
        self.c = c
        self._packs = PackCache()

    @torch.no_grad()
    @ops.range_guarded
    def forward(self, df, f, o, c):
        """AccFlow_.py:97-109.  Concats are laid out as channel slices of shared buffers:
        A = [df | f | o] (2c+1), G = [x | c | f_ | df] (4c), E = [f_ | df | o] (2c+1)."""
        require_cuda(df, f, o, c)
        pk, C = self._packs, self.c
        B, _, h, w = df.shape
        dev = df.device

        def buf(ch):
            return torch.empty((B, ch, h, w), dtype=torch.float32, device=dev)

        A, E, G, X2 = buf(2 * C + 1), buf(2 * C + 1), buf(4 * C), buf(2 * C)
        ops.copy_into(df.float(), A[:, :C]); ops.copy_into(f.float(), A[:, C:2 * C]); ops.copy_into(o.float(), A[:, 2 * C:])
        ops.copy_into(c.float(), X2[:, C:]); ops.copy_into(c.float(), G[:, C:2 * C])
        ops.copy_into(df.float(), G[:, 3 * C:]); ops.copy_into(df.float(), E[:, C:2 * C]); ops.copy_into(o.float(), E[:, 2 * C:])
        t = ops.conv2d(pk.conv("1a", self.conv1[0]), A, act=ops.ACT_RELU)
        ops.conv2d(pk.conv("1b", self.conv1[2]), t, out=X2[:, :C])
        t = ops.conv2d(pk.conv("2a", self.conv2[0]), X2, act=ops.ACT_RELU, out=t)
        u = ops.conv2d(pk.conv("2b", self.conv2[2]), t, act=ops.ACT_RELU)
        zc = self.conv2[4]
        # ZeroConv2d (modules.py:94-96) with exp(3*scale) folded; sigmoid applies to the 9 mask channels only
        om = ops.conv2d(pk.conv("2z", zc.conv, scale=zc.out_scale, scale_dep=zc.scale), u)
        off, msk = om[:, :18], ops.activation_(om[:, 18:], ops.ACT_SIGMOID)  # split [18, 9] (:102-103)
        f_ = ops.conv2d(pk.conv("dc", self.dconv_as_conv(), tap_major=True), A[:, C:2 * C], out=E[:, :C],
                        offset=off, dmask=msk)
        ops.copy_into(f_, G[:, 2 * C:3 * C])
        t = ops.conv2d(pk.conv("3a", self.conv3[0]), E, act=ops.ACT_RELU, out=t)
        ops.conv2d(pk.conv("3b", self.conv3[2]), t, out=G[:, :C])
        t = ops.conv2d(pk.conv("4a", self.conv4[0]), G, act=ops.ACT_RELU, out=t)
        u = ops.conv2d(pk.conv("4b", self.conv4[2]), t, act=ops.ACT_RELU, out=u)
        return ops.conv2d(pk.conv("4c", self.conv4[4]), u)

    def dconv_as_conv(self):
        return _ConvView(self.dconv)


class _ConvView:
    """Presents DeformConv2d's parameters with the attribute names PackCache.conv reads."""

    def __init__(self, d):
        self.weight, self.bias, self.stride, self.padding = d.weight, d.bias, d.stride, d.padding


class Blending(nn.Module):
    def __init__(self, c=128):
        super().__init__()
        self.mask = nn.Sequential(nn.Conv2d(c, c * 2, 1, 1, 0), nn.ReLU(True), nn.Conv2d(c * 2, 1, 3, 1, 1), nn.Sigmoid())
        self._packs = PackCache()

    @torch.no_grad()
    @ops.range_guarded
    def forward(self, f1, f2, emap):
        require_cuda(f1, f2, emap)
        pk = self._packs
        t = ops.conv2d(pk.conv("0", self.mask[0]), emap.float(), act=ops.ACT_RELU)
        m = ops.conv2d(pk.conv("2", self.mask[2]), t, act=ops.ACT_SIGMOID)
        return ops.blend(f1.float().contiguous(), f2.float().contiguous(), m)


def getOcc(F12, I1, I2, binary=True):
    """AccFlow_.py:127-135: 1 where mean_c |I1 - backwarp(I2, F12)| <= 1 (binary) or the per-channel
    absolute error map."""
    require_cuda(F12, I1, I2)
    return ops.get_occ(F12.float(), I1.float(), I2.float(), binary=binary)


def downflow8(flow, mode="bilinear"):
    """AccFlow_.py:138-142."""
    if mode != "bilinear":
        raise NotImplementedError("downflow8: bilinear only")
    require_cuda(flow)
    h, w = flow.shape[-2:]
    assert h % 8 == 0 and w % 8 == 0
    return ops.downflow8(flow.float().contiguous())


class AccFlow(nn.Module):
    """warm_start (build extension, default off = the reference's behaviour): SURVEY 8(f)#2 / the reference README's
    open TODO "Add warmstart mode" on top of RAFT's `flow_init` (raft.py:123-124).  The long-range estimate i -> 0 of
    step i is seeded with the composition of the adjacent flow i -> i-1 and the flow i-1 -> 0 accumulated so far
    (both at 1/8 resolution: seed = F(i->i-1) + backwarp(F(i-1->0), F(i->i-1))) and refined for `warm_iters`
    iterations (default: the estimator's 12).  The adjacent pairs stay one batched call; the long-range pairs become
    sequential with the fusion chain.  oracle.accflow_forward_warm is the CPU restatement."""

    def __init__(self, ofe: nn.Module, warm_start=False, warm_iters=None):
        super().__init__()
        self.warm_start = bool(warm_start)
        self.warm_iters = warm_iters
        self.ofe: nn.Module = ofe
        self.hidden_channel = 128
        self.flow_encoder = FlowEncoder(self.hidden_channel)
        self.flow_decoder = FlowDecoder(self.hidden_channel)
        self.context = BasicEncoder(3, output_dim=self.hidden_channel, norm_fn="none")
        self.accplus = AccPlus(self.hidden_channel)
        self.blending = Blending(self.hidden_channel)
        self.mixed_precision = True

    # ---- one fusion step given all its inputs (AccFlow_.py:191-201) -----------------------------
    def _fuse(self, dflow, flow_ini, F2n, c1, c2, cn):
        f_ini, df, f = self.flow_encoder([flow_ini, dflow, F2n])
        o = getOcc(dflow, c1, c2)
        f_acc = self.accplus(df, f, o, c1)
        emap = getOcc(flow_ini, c1, cn, binary=False)
        f_fuse = self.blending(f_ini, f_acc, emap)
        return self.flow_decoder(f_fuse)

    @torch.no_grad()
    @ops.range_guarded
    def iter(self, I1, I2, In, F2n):
        """input: I1, I2, IN; F2N (1/8 size) -> F1N_small (1/8 size), F1N   (AccFlow_.py:177-201)"""
        require_cuda(I1, I2, In)
        if F2n is None:
            flows = downflow8(self.ofe(torch.cat([I1, I1, I2]), torch.cat([I2, In, In])))
            dflow, flow_ini, F2n = flows.chunk(3)
        else:
            flows = downflow8(self.ofe(torch.cat([I1, I1]), torch.cat([I2, In])))
            dflow, flow_ini = flows.chunk(2)
        c1, c2, cn = self.context([I1, I2, In])
        out_small, out = self._fuse(dflow.contiguous(), flow_ini.contiguous(), F2n.float().contiguous(), c1, c2, cn)
        return out_small.float(), out.float()

    @staticmethod
    def pair_schedule(n_frames):
        """Estimator pairs (i -> j) the reference evaluates for an n-frame sequence, in its order
        (AccFlow_.py:167-190): step 2 -> (2,1),(2,0),(1,0); step i>2 -> (i,i-1),(i,0)."""
        pairs = []
        for i in range(2, n_frames):
            pairs += [(i, i - 1), (i, 0)]
            if i == 2:
                pairs.append((1, 0))
        return pairs

    @torch.no_grad()
    @ops.range_guarded
    def estimate_small(self, images, pairs, flow_init=None, iters=None, features=None):
        """1/8-resolution estimator flows of `pairs`, (len(pairs)*N, 2, H/8, W/8), pair-major."""
        iters = getattr(self, "ofe_iters", 12) if iters is None else iters
        if hasattr(self.ofe, "estimate_pairs"):
            flows = self.ofe.estimate_pairs(images, pairs, iters=iters, flow_init=flow_init, features=features)
        else:  # any estimator with the reference signature
            flows = self.ofe(torch.cat([images[i] for i, _ in pairs]), torch.cat([images[j] for _, j in pairs]),
                             iters=iters, flow_init=flow_init)
        return downflow8(flows)

    @torch.no_grad()
    @ops.range_guarded
    def fuse_chain(self, images, by_pair):
        """The sequential part of AccFlow.forward: by_pair[(i, j)] = (N,2,H/8,W/8) flow i -> j."""
        n = len(images)
        ctx = self.context([im.float().contiguous() for im in images])
        outs, F2n = [], by_pair[(1, 0)]
        for i in range(2, n):
            F2n, up = self._fuse(by_pair[(i, i - 1)].contiguous(), by_pair[(i, 0)].contiguous(), F2n.contiguous(),
                                 ctx[i], ctx[i - 1], ctx[0])
            outs.append(up)
        return outs

    @torch.no_grad()
    @ops.range_guarded
    def forward_warm(self, images):
        """The warm-start schedule (see the class docstring): adjacent pairs in one batch, then per step i the seeded
        estimate i -> 0 followed by the fusion step."""
        images = list(images)
        require_cuda(*images)
        n = len(images)
        if n < 3:
            return []
        N = images[0].shape[0]
        feats = {}
        adjacent = [(i, i - 1) for i in range(1, n)]
        small = self.estimate_small(images, adjacent, features=feats)
        adj = {p: small[k * N:(k + 1) * N] for k, p in enumerate(adjacent)}
        ctx = self.context([im.float().contiguous() for im in images])
        outs, F2n = [], adj[(1, 0)]
        for i in range(2, n):
            dflow = adj[(i, i - 1)].contiguous()
            seed = ops.compose_flow(dflow, F2n.contiguous())
            flow_ini = self.estimate_small(images, [(i, 0)], flow_init=seed, iters=self.warm_iters, features=feats)
            F2n, up = self._fuse(dflow, flow_ini.contiguous(), F2n.contiguous(), ctx[i], ctx[i - 1], ctx[0])
            outs.append(up)
        return outs

    @torch.no_grad()
    @ops.range_guarded
    def forward(self, images, test_mode=False):  # test_mode is ignored by the reference too (:157 FIXME)
        images = list(images)
        require_cuda(*images)
        if len(images) < 3:
            return []
        if self.warm_start:
            return self.forward_warm(images)
        N = images[0].shape[0]
        pairs = self.pair_schedule(len(images))
        small = self.estimate_small(images, pairs)
        return self.fuse_chain(images, {p: small[k * N:(k + 1) * N] for k, p in enumerate(pairs)})

    @torch.no_grad()
    def forward_pair_sharded(self, images, dst=0, group=None):
        """One sequence batch spread over the ranks of `group`: estimator pairs are dealt round-robin, one
        all_gather of the 1/8-res flows, fusion chain on `dst` (returns None on the other ranks)."""
        from ..parallel import run_pair_sharded
        images = list(images)
        N = images[0].shape[0]
        h, w = images[0].shape[2] // 8, images[0].shape[3] // 8

        def est(my_pairs):
            if not my_pairs:
                return torch.zeros((0, N, 2, h, w), dtype=torch.float32, device=images[0].device)
            return self.estimate_small(images, my_pairs).view(len(my_pairs), N, 2, h, w)

        # GMA: the pairs out of one image1 share its attention matrix - keep them on one rank (parallel.deal_pairs)
        shares = hasattr(getattr(self, "ofe", None), "att")
        return run_pair_sharded(est, lambda bp: self.fuse_chain(images, bp), len(images),
                                self.pair_schedule(len(images)), dst=dst, group=group, keep_together=shares)
