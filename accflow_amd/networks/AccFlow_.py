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
from ._packs import PackCache, require_cuda, span
from .modules import ZeroConv2d
from .raft.extractor import BasicEncoder

import contextlib
import threading
import os

USE_S16_CHAIN = os.environ.get("ACCFLOW_S16_CHAIN", "1") == "1"   # (0: the round-3 fusion chain on fp32 activations, A/B)
CONTEXT_SIDE_STREAM = os.environ.get("ACCFLOW_CONTEXT_STREAM", "1") == "1"   # (0: context encoder in the serial section, A/B)
# FlowEncoder of flow_ini / dflow, the occlusion / error maps and the blending mask of ALL fusion steps evaluated in one batch ahead
# of the sequential loop (AccFlow._fuse_chain_hoisted; they do not depend on the accumulated flow).  Measured
# (profiles/r04_ab_chain_hoist.txt): -0.3 ms for one sequence at a time, nothing (+0.1 ms) when the chain runs underneath the
# next sequence's estimator, whose kernels fill the batch-1 steps' idle slots anyway.  "auto": on, except inside
# parallel.SequencePipeline; "1" / "0": always / never (A/B).
USE_CHAIN_HOIST = os.environ.get("ACCFLOW_CHAIN_HOIST", "auto")
_CHAIN_TLS = threading.local()


# The fusion chain of a PIPELINED sequence (another sequence's estimator fills the chip next to it) runs its batch-1
# convolutions without split-K: 21.66 -> 21.43 ms per step (profiles/r06_ab_chain_ksplit.txt); 1: split-K there too (rounds 2-5)
PIPELINE_CHAIN_KSPLIT = os.environ.get("ACCFLOW_PIPELINE_CHAIN_KSPLIT", "0") == "1"


@contextlib.contextmanager
def chain_in_pipeline():
    """Marks the enclosed fuse_chain call as running concurrently with another sequence's estimator (SequencePipeline)."""
    prev = (getattr(_CHAIN_TLS, "pipelined", False), getattr(_CHAIN_TLS, "no_ksplit", False))
    _CHAIN_TLS.pipelined, _CHAIN_TLS.no_ksplit = True, not PIPELINE_CHAIN_KSPLIT
    try:
        yield
    finally:
        _CHAIN_TLS.pipelined, _CHAIN_TLS.no_ksplit = prev


@contextlib.contextmanager
def pipeline_chain_arithmetic():
    """model(images) inside this scope computes the fusion chain with the arithmetic of the pipelined modes (no split-K in its
    batch-1 convolutions: another order of the same fp32 sums) - what SequencePipeline's outputs equal bit for bit."""
    prev = getattr(_CHAIN_TLS, "no_ksplit", False)
    _CHAIN_TLS.no_ksplit = not PIPELINE_CHAIN_KSPLIT
    try:
        yield
    finally:
        _CHAIN_TLS.no_ksplit = prev


def _hoist_now():
    if USE_CHAIN_HOIST == "auto":
        return not getattr(_CHAIN_TLS, "pipelined", False)
    return USE_CHAIN_HOIST == "1" or USE_CHAIN_HOIST is True
USE_CHAIN_DEFER_UP = os.environ.get("ACCFLOW_CHAIN_DEFER_UP", "1") == "1"   # (0: every fusion step upsamples its own flow, A/B)
_CTX_STREAMS = {}


def _context_stream(device):
    """ONE side stream per device for the context encoder / the fusion chain of the pipelined modes: with the caller's stream
    and the estimator's two pair-group streams that makes 4 - the number of hardware queues HIP maps streams onto.  Every
    further stream of the process shares a queue with one of these and serialises with it whenever both are busy (round 6:
    a second pair of group streams cost 6 ms per sequence, profiles/r06_ab_group_priority.txt), so parallel.SequencePipeline
    and forward_pair_sharded_stream use THIS stream for their chains instead of creating their own."""
    key = str(device)
    if key not in _CTX_STREAMS:
        _CTX_STREAMS[key] = torch.cuda.Stream(device=device)
    return _CTX_STREAMS[key]


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
        if ops.s16_active() and USE_S16_CHAIN:
            return self._forward16(ops.to_s16(x.float()))
        t = ops.conv2d(pk.conv("f0", self.flow[0]), x, act=ops.ACT_RELU)
        flow_small = ops.conv2d(pk.conv("f2", self.flow[2]), t)
        t = ops.conv2d(pk.conv("m0", self.mask[0]), x, act=ops.ACT_RELU, out=t)
        mask = ops.conv2d(pk.conv("m2", self.mask[2]), t)  # no 0.25 factor here (AccFlow_.py:42-43)
        return flow_small, ops.convex_upsample(flow_small, mask)


    def _forward16(self, x16):
        """Pre-split form: the first convolutions of the flow and the mask head read the same tensor - ONE 128 -> 512
        convolution (multi-source kernel), its two halves feeding the 2-channel regression (all taps as a 1x1 on the matrix
        cores + tap sum) and the 1x1 mask convolution."""
        pk = self._packs
        B, _, h, w = x16.shape
        c2 = self.flow[0].out_channels
        t16 = ops.S16.empty(B, 2 * c2, h, w, x16.device)
        ops.conv2d_multi(pk.multi_cat("fm0", [self.flow[0], self.mask[0]]), [x16], act=ops.ACT_RELU, out16=t16, fp32_out=False)
        flow_small = ops.conv2d(pk.conv("f2", self.flow[2]), t16.channels(0, c2))
        mask = ops.conv2d_multi(pk.multi("m2m", self.mask[2]), [t16.channels(c2, 2 * c2)])   # no 0.25 factor (AccFlow_.py:42-43)
        return flow_small, ops.convex_upsample(flow_small, mask)

    def flow16(self, x16, out=None):
        """The flow head alone (AccFlow_.py:40): what the NEXT fusion step needs of this one."""
        pk = self._packs
        B, _, h, w = x16.shape
        t16 = ops.S16.empty(B, self.flow[0].out_channels, h, w, x16.device)
        ops.conv2d_multi(pk.multi("f0m", self.flow[0]), [x16], act=ops.ACT_RELU, out16=t16, fp32_out=False)
        return ops.conv2d(pk.conv("f2", self.flow[2]), t16, out=out)

    def upsample16(self, x16, flow_small):
        """The mask head and the convex upsampling (AccFlow_.py:41-45) of a batch of fused features - the part of the
        decoder no later fusion step depends on."""
        pk = self._packs
        B, _, h, w = x16.shape
        t16 = ops.S16.empty(B, self.mask[0].out_channels, h, w, x16.device)
        ops.conv2d_multi(pk.multi("m0m", self.mask[0]), [x16], act=ops.ACT_RELU, out16=t16, fp32_out=False)
        mask = ops.conv2d_multi(pk.multi("m2m", self.mask[2]), [t16])   # no 0.25 factor (AccFlow_.py:42-43)
        return ops.convex_upsample(flow_small, mask)


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
        if ops.s16_active() and USE_S16_CHAIN:
            x = self.encode16(x.float().contiguous())[0]
        else:
            x = ops.conv2d(pk.conv("1", self.conv1), x.float().contiguous(), act=ops.ACT_RELU)
            x = ops.conv2d(pk.conv("2", self.conv2), x, act=ops.ACT_RELU)
            x = ops.conv2d(pk.conv("3", self.conv3), x)
        if is_list:
            x = torch.split(x, batch_dim, dim=0)
        return x

    def encode16(self, flows):
        """(B, 2, h, w) flows -> (fp32 features, ops.S16 features): the 7x7 convolution of the 2-channel flow as a 1x7 one
        over its row-shifted 16-channel stack (ops.flow_from_coords_s16, the form RAFT's convf1 takes), every conv -> conv
        tensor pre-split."""
        pk = self._packs
        B, _, h, w = flows.shape
        dev = flows.device
        c = self.conv1.out_channels
        stack16 = ops.S16.empty(B, 16, h, w, dev)
        ops.flow_from_coords_s16(flows, None, None, stack16, None, 0, is_flow=True)
        a16 = ops.S16.empty(B, c, h, w, dev)
        ops.conv2d(pk.conv("1s", self.conv1, rows_as_channels=True), stack16, act=ops.ACT_RELU, out16=a16, fp32_out=False)
        b16 = ops.S16.empty(B, 2 * c, h, w, dev)
        ops.conv2d_multi(pk.multi("2m", self.conv2), [a16], act=ops.ACT_RELU, out16=b16, fp32_out=False)
        o16 = ops.S16.empty(B, c, h, w, dev)
        out = ops.conv2d_multi(pk.multi("3m", self.conv3), [b16], out16=o16)
        return out, o16


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
        if ops.s16_active() and USE_S16_CHAIN:
            return self.forward16(ops.to_s16(df.float()), f.float().contiguous(), ops.to_s16(f.float()), ops.to_s16(o.float()),
                                  ops.to_s16(c.float()))
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

    def forward16(self, df16, f, f16, o16, c16):
        """AccFlow_.py:97-109 on pre-split tensors: every torch.cat([...], 1) is the source list of ONE multi-source
        convolution (accflow_conv_desc.src: cat[df, f, o], cat[x, c], cat[f_, df, o], cat[x, c, f_, df] - nothing is
        copied), every conv -> conv tensor exists as ops.S16 only.  f is needed in fp32 as well: the deformable
        convolution samples it bilinearly."""
        pk, C = self._packs, self.c
        B, _, h, w = f.shape
        dev = f.device
        R = ops.ACT_RELU

        def s16(ch):
            return ops.S16.empty(B, ch, h, w, dev)

        t16, x16, u16, f_16, y16 = s16(2 * C), s16(C), s16(C), s16(C), s16(C)
        ops.conv2d_multi(pk.multi("1a", self.conv1[0], splits=[C, C, 1]), [df16, f16, o16], act=R, out16=t16, fp32_out=False)
        ops.conv2d_multi(pk.multi("1b", self.conv1[2]), [t16], out16=x16, fp32_out=False)
        ops.conv2d_multi(pk.multi("2a", self.conv2[0], splits=[C, C]), [x16, c16], act=R, out16=t16, fp32_out=False)
        ops.conv2d_multi(pk.multi("2b", self.conv2[2]), [t16], act=R, out16=u16, fp32_out=False)
        zc = self.conv2[4]
        # ZeroConv2d (modules.py:94-96) with exp(3*scale) folded; sigmoid applies to the 9 mask channels only
        om = ops.conv2d_multi(pk.multi("2z", zc.conv, scale=zc.out_scale, scale_dep=zc.scale), [u16])
        # split [18, 9] (:102-103); the sigmoid of the 9 modulation channels is applied by the columns kernel while sampling
        ops.deform_conv2d_s16(pk.conv("dc", self.dconv_as_conv(), tap_major=True), f, om[:, :18], om[:, 18:], f_16,
                              mask_is_logit=True)
        ops.conv2d_multi(pk.multi("3a", self.conv3[0], splits=[C, C, 1]), [f_16, df16, o16], act=R, out16=t16, fp32_out=False)
        ops.conv2d_multi(pk.multi("3b", self.conv3[2]), [t16], out16=y16, fp32_out=False)
        ops.conv2d_multi(pk.multi("4a", self.conv4[0], splits=[C, C, C, C]), [y16, c16, f_16, df16], act=R, out16=t16,
                         fp32_out=False)
        ops.conv2d_multi(pk.multi("4b", self.conv4[2]), [t16], act=R, out16=u16, fp32_out=False)
        return ops.conv2d_multi(pk.multi("4c", self.conv4[4]), [u16])

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

    def mask16(self, emap):
        """The blending mask alone (AccFlow_.py:119-121, S16 path): it depends on the error map only, so AccFlow.fuse_chain
        evaluates it for all steps of a sequence in one batch."""
        pk = self._packs
        e16 = emap if isinstance(emap, ops.S16) else ops.to_s16(emap.float())    # (the chain hands the map over pre-split)
        B, _, h, w = e16.shape
        t16 = ops.S16.empty(B, self.mask[0].out_channels, h, w, e16.device)
        ops.conv2d_multi(pk.multi("0m", self.mask[0]), [e16], act=ops.ACT_RELU, out16=t16, fp32_out=False)
        return ops.conv2d(pk.conv("2", self.mask[2]), t16, act=ops.ACT_SIGMOID)

    @torch.no_grad()
    @ops.range_guarded
    def forward(self, f1, f2, emap):
        require_cuda(f1, f2, emap)
        pk = self._packs
        if ops.s16_active() and USE_S16_CHAIN:
            e16 = ops.to_s16(emap.float())
            B, _, h, w = e16.shape
            t16 = ops.S16.empty(B, self.mask[0].out_channels, h, w, e16.device)
            ops.conv2d_multi(pk.multi("0m", self.mask[0]), [e16], act=ops.ACT_RELU, out16=t16, fp32_out=False)
            t = t16
        else:
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
    def _fuse(self, dflow, flow_ini, F2n, c1, c2, cn, c1_16=None, defer=None):
        """defer = (x16 slice, flow slice): write the fused features (pre-split) and the 1/8-resolution flow there and return
        (flow_small, None) - the caller upsamples all steps in one batch (fuse_chain)."""
        if c1_16 is not None and ops.s16_active() and USE_S16_CHAIN:
            N = dflow.shape[0]
            feats, feats16 = self.flow_encoder.encode16(torch.cat([flow_ini, dflow, F2n], dim=0).float().contiguous())
            f_ini, f = feats[:N], feats[2 * N:]
            h, w = dflow.shape[2:]
            # the occlusion and error maps, the blending mask and the fused features are read by convolutions only: they
            # are written PRE-SPLIT by their producers (round 6: no fp32 tensor + to_s16 pass in between)
            o16 = ops.get_occ(dflow.float(), c1, c2, binary=True, out16=ops.S16.empty(N, 1, h, w, dflow.device))
            f_acc = self.accplus.forward16(feats16.batch(N, 2 * N), f, feats16.batch(2 * N, 3 * N), o16, c1_16)
            e16 = ops.get_occ(flow_ini.float(), c1, cn, binary=False, out16=ops.S16.empty(N, c1.shape[1], h, w, dflow.device))
            m = self.blending.mask16(e16)
            x16 = defer[0] if defer is not None else ops.S16.empty(N, f_acc.shape[1], h, w, dflow.device)
            ops.blend(f_ini.contiguous(), f_acc, m, out16=x16)
            if defer is not None:
                return self.flow_decoder.flow16(x16, out=defer[1]), None
            return self.flow_decoder._forward16(x16)
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
    def estimate_small(self, images, pairs, flow_init=None, iters=None, features=None):
        """1/8-resolution estimator flows of `pairs`, (len(pairs)*N, 2, H/8, W/8), pair-major."""
        iters = getattr(self, "ofe_iters", 12) if iters is None else iters
        if hasattr(self.ofe, "estimate_pairs"):
            flows = self.ofe.estimate_pairs(images, pairs, iters=iters, flow_init=flow_init, features=features)
        else:  # any estimator with the reference signature
            flows = self.ofe(torch.cat([images[i] for i, _ in pairs]), torch.cat([images[j] for _, j in pairs]),
                             iters=iters, flow_init=flow_init)
        return downflow8(flows)

    # ---- the context encoder does not depend on any flow (AccFlow_.py:191: c1, c2, cn = self.context([I1, I2, In])): it runs
    # on a side stream UNDERNEATH the estimator of the same sequence instead of in the serial section behind it
    def context_async(self, images):
        """Start the per-frame context encoding on the side stream; returns a handle for context_join."""
        dev = images[0].device
        main = torch.cuda.current_stream(dev)
        side = _context_stream(dev)
        side.wait_stream(main)          # the images / the weights' packs, and every earlier reader of this pool's blocks
        # the encoder runs concurrently with the estimator's guarded stages, so it reports range violations of the f16x3
        # mode to a flag of its own (read at the join) instead of the thread's shared one
        guarded = ops.current_mode() == ops.CONV_F16X3 and not ops.inside_guard()
        with torch.cuda.stream(side):
            flag = torch.zeros(1, dtype=torch.int32, device=dev) if guarded else None
            with (ops.guard_scope(flag) if guarded else contextlib.nullcontext()):
                ctx = self.context([im.float().contiguous() for im in images], want16=True)
        return (side, ctx, flag, images)

    def context_join(self, handle):
        side, ctx, flag, images = handle
        main = torch.cuda.current_stream(ctx[0][0].device)
        main.wait_stream(side)
        if flag is not None and int(flag.item()):        # the context stage alone falls back to bf16x6 (on this stream)
            ops.note_guard_trip("AccFlow.context")
            with ops.conv_mode(ops.CONV_BF16X6):
                return self.context([im.float().contiguous() for im in images], want16=True)
        # the outputs live in blocks of the side stream's allocator pool and are read on `main`: tell the allocator, so that
        # freeing them cannot hand a block to another side-stream allocation while main's kernels still read it
        for t in ctx[0]:
            t.record_stream(main)
        if ctx[1] is not None:
            for t in ctx[1]:
                t.data.record_stream(main)
        return ctx

    @torch.no_grad()
    @ops.range_guarded
    def fuse_chain(self, images, by_pair, ctx=None):
        """The sequential part of AccFlow.forward: by_pair[(i, j)] = (N,2,H/8,W/8) flow i -> j.  ctx: the context
        encoder's outputs if the caller computed them already (context_async / context_join)."""
        if ctx is None:
            ctx = self.context([im.float().contiguous() for im in images], want16=True)
        if getattr(_CHAIN_TLS, "no_ksplit", False):      # (chain_in_pipeline / pipeline_chain_arithmetic: the chain proper only)
            with ops.ksplit_scope(False):
                return self._fuse_chain(images, by_pair, ctx)
        return self._fuse_chain(images, by_pair, ctx)

    def _fuse_chain(self, images, by_pair, ctx):
        n = len(images)
        ctx, ctx16 = ctx
        outs, F2n = [], by_pair[(1, 0)]
        if ctx16 is not None and ops.s16_active() and USE_S16_CHAIN and _hoist_now() and n > 3:
            return self._fuse_chain_hoisted(n, by_pair, ctx, ctx16)
        defer = ctx16 is not None and ops.s16_active() and USE_S16_CHAIN and USE_CHAIN_DEFER_UP and n > 3
        if defer:
            # Only the 1/8-resolution flow of step i enters step i+1 (AccFlow_.py:171-175): the mask head and the convex
            # upsampling of every step leave the sequential loop and run once, batched over the steps, behind it
            N, _, h, w = F2n.shape
            x16_all = ops.S16.empty((n - 2) * N, self.hidden_channel, h, w, F2n.device)
            small_all = torch.empty(((n - 2) * N, 2, h, w), dtype=torch.float32, device=F2n.device)
        for i in range(2, n):
            k0, k1 = (i - 2) * F2n.shape[0], (i - 1) * F2n.shape[0]
            F2n, up = self._fuse(by_pair[(i, i - 1)].contiguous(), by_pair[(i, 0)].contiguous(), F2n.contiguous(),
                                 ctx[i], ctx[i - 1], ctx[0], c1_16=ctx16[i] if ctx16 is not None else None,
                                 defer=(x16_all.batch(k0, k1), small_all[k0:k1]) if defer else None)
            outs.append(up)
        if defer:
            outs = list(self.flow_decoder.upsample16(x16_all, small_all).split(N, dim=0))
        return outs

    def _fuse_chain_hoisted(self, n, by_pair, ctx, ctx16):
        """fuse_chain with what does not depend on the accumulated flow F2n taken out of the sequential loop and batched over
        the n - 2 steps (AccFlow_.py:191-200 per step: flow_ini, dflow, the context features - hence the occlusion map, the
        error map and the blending mask - are known once the estimator has run): FlowEncoder over [flow_ini, dflow] of all
        steps (one batch of 2 (n-2) N instead of n-2 batches of 2N next to F2n's N), getOcc x 2, Blending.mask ahead of the
        loop; the mask head + convex upsampling behind it (USE_CHAIN_DEFER_UP).  Per step there remain FlowEncoder(F2n),
        AccPlus, the blend and the flow head.  Same operators on the same values, no added work.  (Convolving the df / o / c
        members of AccPlus's concatenations ahead of the loop as partial sums was built in round 4, measured - no gain,
        profiles/r04_ab_chain_prefold.txt - and removed in round 6.)"""
        steps = list(range(2, n))
        N = by_pair[(1, 0)].shape[0]
        K = len(steps) * N
        flow_ini = torch.cat([by_pair[(i, 0)] for i in steps], dim=0).float().contiguous()
        dflow = torch.cat([by_pair[(i, i - 1)] for i in steps], dim=0).float().contiguous()
        feats, feats16 = self.flow_encoder.encode16(torch.cat([flow_ini, dflow], dim=0))
        f_ini, df16 = feats[:K], feats16.batch(K, 2 * K)
        # (views, not copies: the per-frame context features are consecutive slices of ONE encoder output)
        c1 = span([ctx[i] for i in steps])
        c2 = span([ctx[i - 1] for i in steps])
        cn = ctx[0].expand(len(steps), -1, -1, -1) if N == 1 else ctx[0].repeat(len(steps), 1, 1, 1)
        h, w = dflow.shape[2:]
        o16 = ops.get_occ(dflow, c1, c2, binary=True, out16=ops.S16.empty(K, 1, h, w, dflow.device))
        m = self.blending.mask16(ops.get_occ(flow_ini, c1, cn, binary=False, out16=ops.S16.empty(K, c1.shape[1], h, w, dflow.device)))
        outs, F2n = [], by_pair[(1, 0)]
        defer = USE_CHAIN_DEFER_UP
        if defer:
            _, _, h, w = F2n.shape
            x16_all = ops.S16.empty(K, self.hidden_channel, h, w, F2n.device)
            small_all = torch.empty((K, 2, h, w), dtype=torch.float32, device=F2n.device)
        for k, i in enumerate(steps):
            k0, k1 = k * N, (k + 1) * N
            f, f16 = self.flow_encoder.encode16(F2n.float().contiguous())
            f_acc = self.accplus.forward16(df16.batch(k0, k1), f, f16, o16.batch(k0, k1), ctx16[i])
            if defer:
                F2n = self.flow_decoder.flow16(ops.blend(f_ini[k0:k1], f_acc, m[k0:k1], out16=x16_all.batch(k0, k1)),
                                               out=small_all[k0:k1])
            else:
                F2n, up = self.flow_decoder(ops.blend(f_ini[k0:k1], f_acc, m[k0:k1]))
                outs.append(up)
        if defer:
            outs = list(self.flow_decoder.upsample16(x16_all, small_all).split(N, dim=0))
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

    @staticmethod
    def stack_frames(images):
        """The frames as consecutive views of ONE tensor (a single copy): the three encoders then take their frame lists
        without a torch.cat each (_packs.span; 41 MB per 7-frame 480x1024 list)."""
        images = [im.float() for im in images]
        if len({tuple(im.shape) for im in images}) != 1:
            return images
        whole = span([im.contiguous() for im in images])      # (already adjacent: no copy at all)
        return list(whole.split(images[0].shape[0], dim=0))

    @torch.no_grad()
    def forward(self, images, test_mode=False):  # test_mode is ignored by the reference too (:157 FIXME)
        """f16x3 mode: the stages - estimator encoders, refinement, context encoder, fusion chain - are range-guarded one by
        one (ops.with_range_guard): a value outside the fp16 split's range re-runs only the stage that saw it in bf16x6."""
        images = list(images)
        require_cuda(*images)
        if len(images) < 3:
            return []
        if self.warm_start:
            return self.forward_warm(images)
        images = self.stack_frames(images)
        N = images[0].shape[0]
        pairs = self.pair_schedule(len(images))

        def run():
            handle = self.context_async(images) if CONTEXT_SIDE_STREAM else None
            small = self.estimate_small(images, pairs)
            ctx = self.context_join(handle) if handle is not None else None
            return self.fuse_chain(images, {p: small[k * N:(k + 1) * N] for k, p in enumerate(pairs)}, ctx=ctx)

        # first without a host synchronisation per stage; only if some value left the fp16 split's range, stage by stage
        out, tripped = ops.optimistic(run, images[0].device)
        return run() if tripped else out

    @torch.no_grad()
    def forward_pair_sharded_stream(self, sequences, group=None, on_result=None):
        """A stream of sequence batches, each spread over all ranks of `group`, with a ROTATING root
        (parallel.run_pair_sharded_stream): sequence k's fusion chain runs on rank k % world - on a side stream, underneath
        that rank's estimator pairs of the following sequences - so every rank does the same number of pairs and chains.
        Returns {k: outputs of sequence k} for the sequences this rank was the root of.  `sequences` may be any iterable (a
        data loader: it is consumed one sequence at a time); pending chains are resolved with a bounded lag, and with
        on_result(k, outputs) nothing is retained here - memory does not grow with the stream's length."""
        import itertools
        from .. import ops
        from ..parallel import run_pair_sharded_stream
        it = iter(sequences)
        first = list(next(it))
        sequences = itertools.chain([first], (list(s) for s in it))
        dev, n_frames = first[0].device, len(first)
        if getattr(self, "_stream_side", None) is None:
            self._stream_side = _context_stream(dev)      # (ONE side stream per device in the process: hardware queues, see there)
        side = self._stream_side
        guarded = ops.current_mode() == ops.CONV_F16X3

        # f16x3 range guard without a host synchronisation: every rank's estimator pairs report to a device flag that rides
        # in the sequence's all_gather (parallel.run_pair_sharded_stream's aux); the root adds its chain's flag and copies
        # the sum to pinned memory behind the chain.  A tripped sequence (no real image has produced one) is recomputed
        # by its root alone, all pairs, in bf16x6, at the harvest.
        def est(images, my_pairs, is_root):
            N = images[0].shape[0]
            h, w = images[0].shape[2] // 8, images[0].shape[3] // 8
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
            if not my_pairs:
                return torch.zeros((0, N, 2, h, w), dtype=torch.float32, device=dev), flag
            with (ops.guard_scope(flag) if guarded else contextlib.nullcontext()):
                small = self.estimate_small(images, my_pairs).view(len(my_pairs), N, 2, h, w)
            return small, flag

        def fuse(images, bp, flags):
            main = torch.cuda.current_stream(dev)
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
            ready = torch.cuda.Event()
            ready.record(main)
            side.wait_event(ready)
            host = None
            with torch.cuda.stream(side), chain_in_pipeline(), (ops.guard_scope(flag) if guarded else contextlib.nullcontext()):
                outs = self.fuse_chain(images, bp)
                if guarded:
                    tripped = flag + (torch.cat([f.reshape(-1) for f in flags]) != 0).sum().to(torch.int32)
                    host = torch.empty(1, dtype=torch.int32, pin_memory=True)
                    host.copy_(tripped, non_blocking=True)
                done = torch.cuda.Event()
                done.record(side)
            for o in outs:
                o.record_stream(main)
            return (done, host, outs, (images, bp, flag, flags))

        def harvest(h):
            done, host, outs, keep = h
            done.synchronize()
            if host is not None and int(host.item()):      # a value left the fp16 split's range: this sequence again in bf16x6
                ops.note_guard_trip("AccFlow.forward_pair_sharded_stream")
                with ops.conv_mode(ops.CONV_BF16X6), pipeline_chain_arithmetic():    # (every output of the mode: one arithmetic)
                    outs = self(images=keep[0])
            return outs

        shares = hasattr(getattr(self, "ofe", None), "att")
        return run_pair_sharded_stream(est, fuse, self.pair_schedule(n_frames), sequences, group=group,
                                       keep_together=shares, harvest=harvest, on_result=on_result)

    @torch.no_grad()
    def forward_pair_sharded(self, images, dst=0, group=None):
        """One sequence batch spread over the ranks of `group`: estimator pairs are dealt round-robin, one
        all_gather of the 1/8-res flows, fusion chain on `dst` (returns None on the other ranks)."""
        from ..parallel import run_pair_sharded
        images = list(images)
        N = images[0].shape[0]
        h, w = images[0].shape[2] // 8, images[0].shape[3] // 8

        from ..parallel import world
        is_root = world(group)[1] == dst
        holder = []

        def est(my_pairs):
            # the root encodes the context features of all frames on its side stream WHILE it estimates its own pairs:
            # the serial section behind the all_gather is then the five fusion steps only
            if is_root and CONTEXT_SIDE_STREAM and images[0].is_cuda and hasattr(self, "context_async") and not holder:
                holder.append(self.context_async(images))
            if not my_pairs:
                return torch.zeros((0, N, 2, h, w), dtype=torch.float32, device=images[0].device)
            return self.estimate_small(images, my_pairs).view(len(my_pairs), N, 2, h, w)

        def fuse(bp):
            if holder:
                return self.fuse_chain(images, bp, ctx=self.context_join(holder[0]))
            return self.fuse_chain(images, bp)

        # GMA: the pairs out of one image1 share its attention matrix - keep them on one rank (parallel.deal_pairs)
        shares = hasattr(getattr(self, "ofe", None), "att")
        return run_pair_sharded(est, fuse, len(images), self.pair_schedule(len(images)), dst=dst, group=group,
                                keep_together=shares)
