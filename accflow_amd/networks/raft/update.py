"""RAFT update block (motion encoder -> SepConvGRU -> flow / mask heads) on fused HIP conv kernels.

Parameter names follow the reference's networks/raft/update.py (FlowHead :6-14, SepConvGRU :33-60,
BasicMotionEncoder :80-97, BasicUpdateBlock :112-136) so checkpoints load unchanged.  The data flow is
re-organised around one persistent channel-sliced buffer per pair batch so that no torch.cat and no
stand-alone gate arithmetic is ever executed:

    HX  (B, 128+X, h, w) = [ h | inp | motion(126) , flow(2) | (GMA: motion_global) ]
    z,r : ONE conv with 256 output channels over HX; epilogue writes z and r*h          (update.py:47-49)
    q   : conv over cat[r*h, x] expressed as two sources; epilogue h <- (1-z)h + z*tanh  (update.py:50-51)
    delta: flow-head conv2 accumulates straight into coords1                             (raft.py:136)

In the f16x3 mode (ops.s16_active()) the tensors that only convolutions read - lookup output, cor / flo / cor_flo, r*h, the
convolution-side copy of h, the flow head's hidden layer - live pre-split as ops.S16 (UpdateWorkspace), and the launches of
iterations 2..12 are replayed from descriptors kept on the workspace.
"""
import torch
import torch.nn as nn

import os

from ... import ops
from .._packs import PackCache, require_cuda

# The CorrBlock lookup fused with the motion encoder's first convolution (csrc/corr_lookup_conv.hip): the 4 x 81 taps never
# reach HBM.  ACCFLOW_FUSE_LOOKUP=0 runs the two launches (S16 lookup, then convc1) of round 3.
FUSE_LOOKUP = os.environ.get("ACCFLOW_FUSE_LOOKUP", "1") == "1"
# the GRU state h as the pre-split tensor only inside the refinement loop (0: an fp32 copy beside it, rounds 3-5, A/B)
USE_H16_STATE = os.environ.get("ACCFLOW_H16_STATE", "1") == "1"


def h16_state_supported():
    """The pre-split-only state rides on the tap-specialised 5-tap kernels of the direct route (csrc/conv2d.hip)."""
    return (not ops.S16_VIA_MULTI and os.environ.get("ACCFLOW_DIRECT_KT", "1") != "0" and os.environ.get("ACCFLOW_DIRECT_W4", "1") != "0"
            and os.environ.get("ACCFLOW_S16M", "0") == "0")


class FlowHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, 2, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)


class SepConvGRU(nn.Module):
    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        cin = hidden_dim + input_dim
        for suffix, ks, pad in (("1", (1, 5), (0, 2)), ("2", (5, 1), (2, 0))):
            for gate in "zrq":
                setattr(self, "conv%s%s" % (gate, suffix), nn.Conv2d(cin, hidden_dim, ks, padding=pad))
        self.hidden_dim = hidden_dim
        self.input_dim = input_dim


class BasicMotionEncoder(nn.Module):
    def __init__(self, args):
        super().__init__()
        cor_planes = args.corr_levels * (2 * args.corr_radius + 1) ** 2
        self.convc1 = nn.Conv2d(cor_planes, 256, 1, padding=0)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 192, 128 - 2, 3, padding=1)


class UpdateWorkspace:
    """Device buffers of one pair batch for the iterative refinement (all (B, C, h, w) fp32)."""

    def __init__(self, B, h, w, device, hidden=128, x_dim=256):
        def buf(c):
            return torch.empty((B, c, h, w), dtype=torch.float32, device=device)
        self.B, self.h, self.w, self.hidden, self.x_dim = B, h, w, hidden, x_dim
        self._buf = buf
        self.s16 = bool(ops.s16_active())
        self.net_in_h16 = False
        self.c1_fused = False   # set by the caller that ran relu(convc1(lookup)) as ONE kernel into c1_16
        self.descs = {}    # filled conv descriptors of the iteration's call sites (ops.conv2d cache=)
        self.hx = buf(hidden + x_dim)
        self.net = self.hx[:, :hidden]                      # h
        self.inp = self.hx[:, hidden:hidden + 128]          # context features
        self.motion = self.hx[:, hidden + 128:hidden + 256]  # [conv out 126 | flow 2]
        self.motion_conv = self.hx[:, hidden + 128:hidden + 254]
        self.motion_flow = self.hx[:, hidden + 254:hidden + 256]
        self.x = self.hx[:, hidden:]
        self.z = buf(hidden)
        self.flow = buf(2)
        self._corr = None
        if not self.s16:    # fp32 conv-to-conv activations (in S16 mode their pre-split forms below replace them)
            self.rh = buf(hidden)
            self.flow16 = buf(16)                           # row-shifted stack of the flow (input of convf1 as a 1x7 conv)
            self.c1 = buf(256)
            self.corflo = buf(256)
            self.f1 = buf(128)
            self.head = buf(256)
        self.mask = None
        self.gru_pre = None   # W[:, inp] * inp of the four GRU gate convs (BasicUpdateBlock.gru_context)
        # S16 mode (ops.s16_active()): every conv-to-conv activation of an iteration lives PRE-SPLIT (ops.S16) - the
        # consumer's patch loader is then a DMA, the producer's epilogue does the fp16 split.  fp32 copies remain only
        # where something other than a convolution reads them: h (the GRU state: epilogue operand e0, final upsampling
        # mask input comes from h16), z, the 2-channel flow, coords.
        if self.s16:
            def s(c, zero=False):
                return ops.S16.empty(B, c, h, w, device, zero=zero)
            self._corr16 = None                # (allocated on first use: the fused lookup -> convc1 kernel never writes it)
            self.stack16 = s(16)
            self.c1_16, self.corflo16, self.f1_16, self.head16 = s(256), s(256), s(128), s(256)
            self.h16, self.rh16 = s(hidden), s(hidden)
            # round 6: inside the refinement loop the GRU epilogues run on PACKED operands (accflow_conv_desc.e0_fmt / p32): the
            # state lives in h16 ONLY (h = (hi + lo) / 2^4 read with 8-byte loads; the q launches write no fp32 copy, 43 MB
            # each at B = 11), z and the context addends (gru_pre) are PIXEL-MAJOR fp32 tensors (one 16-byte load per 4
            # channels); self.net is then valid only until the first gru_step (net32() gives the current state) and self.z /
            # gru_pre hold the pixel-major layout (ops.from_p32)
            self.net_in_h16 = USE_H16_STATE and h16_state_supported()
            self.x16 = s(x_dim - 128)          # the GRU input without the context features: [motion | (GMA: motion_global)]
            self.motion16 = self.x16.channels(0, 128)


def _fill_s16_inputs(ws, flow_or_coords, is_flow):
    """Module-boundary entry (the reference's BasicUpdateBlock.forward signature hands over fp32 tensors): convert what
    the S16 iteration expects - the correlation features in the S16 lookup order, h, and the flow pieces."""
    if not ws.s16:
        return
    B, h, w = ws.B, ws.h, ws.w
    c = ws.corr.view(B, 4, 9, 9, h, w).transpose(2, 3).reshape(B, 4, 81, h, w)   # [l][i][j] -> [l][j][i]
    c88 = torch.zeros((B, 4, 88, h, w), dtype=torch.float32, device=ws.corr.device)
    c88[:, :, :81] = c
    ops.to_s16(c88.view(B, 352, h, w), ws.corr16)
    ops.to_s16(ws.net, ws.h16)
    ops.flow_from_coords_s16(flow_or_coords, ws.flow, ws.motion_flow, ws.stack16, ws.motion16, 126, is_flow=is_flow)


UpdateWorkspace.fill_s16_inputs = _fill_s16_inputs


def _ws_corr(ws):
    """(B, 324, h, w) fp32 lookup output: the iteration's buffer in fp32 mode; in S16 mode only module-boundary calls
    (the reference's forward signatures) touch it, so it is allocated on first use."""
    if ws._corr is None:
        ws._corr = ws._buf(324)
    return ws._corr


UpdateWorkspace.corr = property(_ws_corr)


def _ws_corr16(ws):
    """(B, 4 x 88, h, w) pre-split lookup output: only the two-launch form (ACCFLOW_FUSE_LOOKUP=0, module-boundary calls,
    bench.py's stand-alone lookup timing) writes it."""
    if ws._corr16 is None:
        ws._corr16 = ops.S16.empty(ws.B, ops.LOOKUP_S16_CHANNELS, ws.h, ws.w, ws.hx.device)
    return ws._corr16


UpdateWorkspace.corr16 = property(_ws_corr16)


def _ws_net32(ws):
    """The GRU state as an fp32 tensor (module-boundary calls): de-split from h16 when the loop keeps it there only."""
    return ws.h16.to_float() if (ws.s16 and ws.net_in_h16) else ws.net.contiguous()


UpdateWorkspace.net32 = _ws_net32


class BasicUpdateBlock(nn.Module):
    def __init__(self, args, hidden_dim=128, input_dim=128):
        super().__init__()
        self.args = args
        self.encoder = BasicMotionEncoder(args)
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=256)
        self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(256, 64 * 9, 1, padding=0))
        self._packs = PackCache()

    def prepack(self):
        """Build every weight pack of the block on the CURRENT stream.  RAFT._refine calls this before it forks the
        pair groups onto side streams: packs are made lazily on first use, and a pack built inside one group's stream
        would be read by the other group's kernels with no ordering between the two streams."""
        pk, e, g, f = self._packs, self.encoder, self.gru, self.flow_head
        pk.conv("c1", e.convc1); pk.conv("c2", e.convc2); pk.conv("f1s", e.convf1, rows_as_channels=True)
        if ops.s16_active():
            pk.conv("c1s", e.convc1, lookup88=True)
            if FUSE_LOOKUP:
                pk.conv("c1f", e.convc1, lookup_fused=True)
        pk.conv("f2", e.convf2)
        pk.conv("cf", e.conv)
        for s in ("1", "2"):
            self._gru_packs(s)
        pk.conv("fh1", f.conv1); pk.conv("fh2", f.conv2)
        pk.conv("m0", self.mask[0]); pk.conv("m2", self.mask[2], const_scale=0.25)

    # ---- pieces shared with GMAUpdateBlock -------------------------------------------------
    def motion_encoder(self, ws):
        """BasicMotionEncoder.forward (update.py:89-97): corr, flow -> ws.motion_conv (flow slice is
        written by flow_from_coords)."""
        pk, e = self._packs, self.encoder
        if ws.s16:
            R, D = ops.ACT_RELU, ws.descs
            fused = ws.c1_fused    # convc1 already ran inside the lookup kernel (RAFT._lookup_and_flow)
            if "c2" in D:     # iterations 2..: the same launches on the same buffers (descriptors kept on the workspace)
                for k in ("c2", "f1", "f2", "cf") if fused else ("c1", "c2", "f1", "f2", "cf"):
                    ops.conv2d(None, None, cache=(D, k))
                return
            if not fused:
                ops.conv2d(pk.conv("c1s", e.convc1, lookup88=True), ws.corr16, out16=ws.c1_16, act=R, fp32_out=False,
                           algo_cin=324, cache=(D, "c1"))
            ops.conv2d(pk.conv("c2", e.convc2), ws.c1_16, out16=ws.corflo16.channels(0, 192), act=R, fp32_out=False,
                       cache=(D, "c2"))
            ops.conv2d(pk.conv("f1s", e.convf1, rows_as_channels=True), ws.stack16, out16=ws.f1_16, act=R, fp32_out=False,
                       algo_cin=2 * 7, cache=(D, "f1"))
            ops.conv2d(pk.conv("f2", e.convf2), ws.f1_16, out16=ws.corflo16.channels(192, 256), act=R, fp32_out=False,
                       cache=(D, "f2"))
            # 126 channels: the last pair of the octet (the flow, update.py:96) was written by flow_from_coords_s16;
            # GMA's aggregator also reads the motion features in fp32
            want32 = ws.x_dim > 256
            ops.conv2d(pk.conv("cf", e.conv), ws.corflo16, out16=ws.motion16.channels(0, 126), act=R, fp32_out=want32,
                       out=ws.motion_conv if want32 else None, cache=(D, "cf"))
            return
        ops.conv2d(pk.conv("c1", e.convc1), ws.corr, out=ws.c1, act=ops.ACT_RELU)
        ops.conv2d(pk.conv("c2", e.convc2), ws.c1, out=ws.corflo[:, :192], act=ops.ACT_RELU)
        # convf1 (7x7 over the 2-channel flow) as a 1x7 convolution of the 16-channel row-shifted stack: same products
        ops.conv2d(pk.conv("f1s", e.convf1, rows_as_channels=True), ws.flow16, out=ws.f1, act=ops.ACT_RELU, algo_cin=2 * 7)
        ops.conv2d(pk.conv("f2", e.convf2), ws.f1, out=ws.corflo[:, 192:], act=ops.ACT_RELU)
        ops.conv2d(pk.conv("cf", e.conv), ws.corflo, out=ws.motion_conv, act=ops.ACT_RELU)

    def _gru_packs(self, s):
        """The four packs of GRU half-step s.  Every gate conv runs over cat[h, x] with x = [inp | motion ...]
        (update.py:46-50), and `inp` - the context features - does not change over the refinement iterations
        (raft.py:117-133), so by linearity each conv is cut along its INPUT channels:
            W * [h | inp | rest] = W[:, h | rest] * [h | rest]   (every iteration, K smaller by a third)
                                 + W[:, inp] * inp               (once per pair: gru_context)
        -> (zr over [h | rest], zr over inp, q over [r*h | rest], q over inp)."""
        pk, g, hd = self._packs, self.gru, self.gru.hidden_dim
        cin = hd + g.input_dim
        var, ctx = [(0, hd), (hd + 128, cin)], [(hd, hd + 128)]
        zr = [getattr(g, "convz" + s), getattr(g, "convr" + s)]
        q = [getattr(g, "convq" + s)]
        return (pk.conv_cat("zr%sv" % s, zr, C0=hd, in_slices=var), pk.conv_cat("zr%sc" % s, zr, in_slices=ctx, with_bias=False),
                pk.conv_cat("q%sv" % s, q, C0=hd, in_slices=var), pk.conv_cat("q%sc" % s, q, in_slices=ctx, with_bias=False))

    def gru_context(self, ws):
        """The iteration-invariant part of the four gate convolutions: W[:, inp] * inp, once per pair batch."""
        B, h, w, hd = ws.B, ws.h, ws.w, ws.hidden
        ws.gru_pre = {}
        # (S16 mode, round 6: the context features are split ONCE and the four convolutions stage them by DMA on the
        # tap-specialised loop instead of gathering + splitting them per tile)
        inp = ops.to_s16(ws.inp) if ws.s16 else ws.inp
        p32 = ws.s16 and ws.net_in_h16        # (the packed-operand GRU epilogues read the addend pixel-major)
        for s in ("1", "2"):
            _, zrc, _, qc = self._gru_packs(s)
            ws.gru_pre["zr" + s] = ops.conv2d(zrc, inp, algo_cin=0, p32_out=p32)   # (accounted with the per-iteration convs)
            ws.gru_pre["q" + s] = ops.conv2d(qc, inp, algo_cin=0, p32_out=p32)

    def gru_step(self, ws):
        """SepConvGRU.forward (update.py:45-60): two half-steps, h updated in place in ws.hx."""
        if getattr(ws, "gru_pre", None) is None:
            self.gru_context(ws)
        rest = ws.hx[:, ws.hidden + 128:]   # x without the context features: [motion | (GMA: motion_global)]
        cin = ws.hidden + ws.x_dim          # input channels of the gate convs as the reference runs them
        if ws.s16:
            D = ws.descs
            if "zr1" in D:
                for k in ("zr1", "q1", "zr2", "q2"):
                    ops.conv2d(None, None, cache=(D, k))
                return
            h16s = ws.net_in_h16
            for s in ("1", "2"):
                zrv, _, qv, _ = self._gru_packs(s)
                ops.conv2d(zrv, ws.h16, in1=ws.x16, out=ws.z, act=ops.ACT_SIGMOID, epi=ops.EPI_GRU_ZR, e0=ws.h16 if h16s else ws.net,
                           out16=ws.rh16, fp32_out=False, pre=ws.gru_pre["zr" + s], algo_cin=cin, cache=(D, "zr" + s))
                ops.conv2d(qv, ws.rh16, in1=ws.x16, out=None if h16s else ws.net, act=ops.ACT_TANH, epi=ops.EPI_GRU_Q,
                           e0=ws.h16 if h16s else ws.net, e1=ws.z, out16=ws.h16, fp32_out=not h16s,
                           pre=ws.gru_pre["q" + s], algo_cin=cin, cache=(D, "q" + s))
            return
        for s in ("1", "2"):
            zrv, _, qv, _ = self._gru_packs(s)
            ops.conv2d(zrv, ws.net, in1=rest, out=ws.z, act=ops.ACT_SIGMOID, epi=ops.EPI_GRU_ZR, e0=ws.net, out2=ws.rh,
                       pre=ws.gru_pre["zr" + s], algo_cin=cin)
            ops.conv2d(qv, ws.rh, in1=rest, out=ws.net, act=ops.ACT_TANH, epi=ops.EPI_GRU_Q, e0=ws.net, e1=ws.z,
                       pre=ws.gru_pre["q" + s], algo_cin=cin)

    def flow_delta(self, ws, coords1=None, out=None):
        """FlowHead (update.py:13-14).  With coords1 the delta is accumulated in place (raft.py:136)."""
        pk, f = self._packs, self.flow_head
        if ws.s16:
            D = ws.descs
            if coords1 is not None and ops.profiler.ACTIVE is None and "fh12" in D and D["fh12"][1] is coords1:   # replay both launches
                return ops.conv2d_tapgemm(None, None, None, cache=(D, "fh12"))
            p1, p2 = pk.conv("fh1", f.conv1), pk.conv("fh2", f.conv2)
            if ops.tapgemm_eligible(p1, p2, ws.h16):
                # conv1's 256 channels never reach HBM: its epilogue multiplies them by conv2's 18-row tap matrix
                if coords1 is not None:
                    return ops.conv2d_tapgemm(p1, p2, ws.h16, out=coords1, epi=ops.EPI_ACCUM, e0=coords1, cache=(D, "fh12"))
                return ops.conv2d_tapgemm(p1, p2, ws.h16, out=out)
            ops.conv2d(p1, ws.h16, out16=ws.head16, act=ops.ACT_RELU, fp32_out=False, cache=(ws.descs, "fh1"))
            if coords1 is not None:
                return ops.conv2d(pk.conv("fh2", f.conv2), ws.head16, out=coords1, epi=ops.EPI_ACCUM, e0=coords1)
            return ops.conv2d(pk.conv("fh2", f.conv2), ws.head16, out=out)
        ops.conv2d(pk.conv("fh1", f.conv1), ws.net, out=ws.head, act=ops.ACT_RELU)
        if coords1 is not None:
            return ops.conv2d(pk.conv("fh2", f.conv2), ws.head, out=coords1, epi=ops.EPI_ACCUM, e0=coords1)
        return ops.conv2d(pk.conv("fh2", f.conv2), ws.head, out=out)

    def up_mask(self, ws):
        """mask = .25 * self.mask(net) (update.py:135); 0.25 is folded into the packed 1x1 weights."""
        pk = self._packs
        if ws.mask is None:
            ws.mask = torch.empty((ws.B, 576, ws.h, ws.w), dtype=torch.float32, device=ws.hx.device)
        if ws.s16:
            ops.conv2d(pk.conv("m0", self.mask[0]), ws.h16, out16=ws.head16, act=ops.ACT_RELU, fp32_out=False)
            return ops.conv2d(pk.conv("m2", self.mask[2], const_scale=0.25), ws.head16, out=ws.mask)
        ops.conv2d(pk.conv("m0", self.mask[0]), ws.net, out=ws.head, act=ops.ACT_RELU)
        return ops.conv2d(pk.conv("m2", self.mask[2], const_scale=0.25), ws.head, out=ws.mask)

    def step(self, ws, coords1, want_mask):
        """One refinement iteration on the workspace: expects ws.corr and ws.flow / ws.motion_flow set."""
        self.motion_encoder(ws)
        self.gru_step(ws)
        self.flow_delta(ws, coords1=coords1)
        return self.up_mask(ws) if want_mask else None

    @torch.no_grad()
    @ops.range_guarded
    def forward(self, net, inp, corr, flow, upsample=True):
        """Reference signature (update.py:127-136): returns (net, mask, delta_flow)."""
        require_cuda(net, inp, corr, flow)
        B, _, h, w = net.shape
        ws = UpdateWorkspace(B, h, w, net.device)
        ops.copy_into(net.float(), ws.net)
        ops.copy_into(inp.float(), ws.inp)
        ops.copy_into(corr.float(), ws.corr)
        ws.fill_s16_inputs(flow.float().contiguous(), is_flow=True)
        if not ws.s16:
            ops.flow_from_coords(flow.float().contiguous(), dst0=ws.flow, dst1=ws.motion_flow, stack16=ws.flow16, is_flow=True)
        self.motion_encoder(ws)
        self.gru_step(ws)
        delta = self.flow_delta(ws)
        mask = self.up_mask(ws)
        return ws.net32(), mask, delta
