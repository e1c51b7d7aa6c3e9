"""One training step of AccFlow on the HIP library (reference: train_acc.py:203-234, loss.py:30-36,
AccFlow_.py:157-201).

What carries gradients in the reference's step - and therefore here:
  * the optical-flow estimator is frozen (train_acc.py:163-164) and runs under no_grad (AccFlow_.py:183): inference path;
  * the accumulated flow is detached between the steps of a sequence (AccFlow_.py:171-172), the occlusion and error maps
    are detached (AccFlow_.py:195,198): every fusion step is its own graph, tied to the others only through the parameters;
  * inside a step: FlowEncoder (3 convolutions), the context encoder of I1 (norm "none": convolutions + residual adds),
    AccPlus (11 convolutions, ZeroConv2d's exp(3 scale), sigmoid, the modulated deformable convolution), Blending (2
    convolutions, sigmoid, the blend), FlowDecoder (4 convolutions, convex upsampling); loss = sum_i mean |F_i - gt_i|.

Forward runs the same HIP kernels as inference through a small tape (Tape) that keeps the activations a backward needs and
a closure per operator; backward runs the closures in reverse (accflow_amd/backward.py).  Convolutions use the
fp32-equivalent bf16x6 arithmetic (no range condition to guard), fp32 everywhere else - the reference's autocast (fp16)
is a lower-precision approximation of the same graph.  Parameter gradients land in `param.grad` like autograd's, so that
torch.optim.AdamW / clip_grad_norm_ / the OneCycle schedule of train_acc.py:72-87,230-234 apply unchanged (optimizer
arithmetic on parameter-sized tensors is plumbing, not the hot path).

Parity: tests/test_train.py compares every parameter gradient of one 2-step sequence with the reference's own autograd
(tests/golden/accflow_grad_c1.npz, made by tests/golden/make_grad_golden.py from /root/reference in fp32)."""
import contextlib
import os

import torch

from . import backward as B
from . import ops

# Arithmetic of the heads' FORWARD convolutions on the tape.  "f16x3" (default since round 5): the inference path's fp16
# hi + lo split - 3 MFMAs per product, 22-bit operands, fp32 accumulation - with its range guard: the whole step reports to
# one flag, and a step in which a value left the scaled fp16 range is redone in "bf16x6" (6 MFMAs per product, no range
# condition; ACCFLOW_TRAIN_CONV_MODE=bf16x6 makes that the only mode, as in round 4).  The backward convolutions always run
# bf16x6 (backward.GRAD_CONV_MODE): gradients span the whole fp32 exponent range.
TRAIN_CONV_MODE = os.environ.get("ACCFLOW_TRAIN_CONV_MODE", "f16x3")
# The backward of fusion step k runs on a side stream UNDERNEATH the forward of step k+1 (which needs step k's 1/8-resolution
# flow, not its gradients).  Measured: 56.0 -> 54.1 ms per step (tools/train_bench.py, ACCFLOW_TRAIN_OVERLAP=0|1).
OVERLAP_BACKWARD = os.environ.get("ACCFLOW_TRAIN_OVERLAP", "1") == "1"
# The backward passes of the S fusion steps of a sequence are independent given the saved activations: ONE backward over the
# step-concatenated batch (Tape's step mode) instead of S.  ACCFLOW_TRAIN_BATCHED_BW=0: one tape and one backward per step.
BATCHED_BACKWARD = os.environ.get("ACCFLOW_TRAIN_BATCHED_BW", "1") == "1"
_BW_STREAMS = {}


# Parameter gradients are leaves of the backward pass: with ACCFLOW_TRAIN_WGRAD_STREAM=1 the weight-gradient GEMMs run on a
# stream of their own, behind the activation gradient they read, while the input-gradient chain continues on the current one.
# Measured: no gain (55.0 vs 54.8 ms per step) - off by default.
WGRAD_STREAM = os.environ.get("ACCFLOW_TRAIN_WGRAD_STREAM", "0") == "1"
_WG_STREAMS = {}


@contextlib.contextmanager
def _param_grad_stream(*tensors):
    """Run the enclosed parameter-gradient work on the weight-gradient stream, ordered behind everything already enqueued on
    the current stream; `tensors` (read there, allocated here) are marked for the allocator."""
    ts = [t for t in tensors if t is not None]
    if not WGRAD_STREAM or not ts or not ts[0].is_cuda:
        yield
        return
    dev = ts[0].device
    key = str(dev)
    if key not in _WG_STREAMS:
        _WG_STREAMS[key] = torch.cuda.Stream(device=dev)
    wg = _WG_STREAMS[key]
    wg.wait_stream(torch.cuda.current_stream(dev))
    for t in ts:
        t.record_stream(wg)
    with torch.cuda.stream(wg):
        yield


def join_param_grads(device):
    """Order the current stream behind the weight-gradient stream (before anything reads `param.grad`)."""
    wg = _WG_STREAMS.get(str(device))
    if wg is not None:
        torch.cuda.current_stream(device).wait_stream(wg)


def _backward_stream(device):
    key = str(device)
    if key not in _BW_STREAMS:
        _BW_STREAMS[key] = torch.cuda.Stream(device=device)
    return _BW_STREAMS[key]


class Var:
    """A tensor on the tape: value, gradient (None until a consumer's backward delivers one), and whether anything
    upstream wants it."""
    __slots__ = ("v", "g", "needs")

    def __init__(self, v, needs=True):
        self.v, self.g, self.needs = v, None, needs

    def acc(self, g):
        if not self.needs:
            return
        if self.g is None:
            self.g = g
        else:
            B.add_(self.g, g)


class Tape:
    """Operators record a backward closure.  STEP MODE (begin_steps / step / end_steps): the same operator sequence is run
    S times on consecutive batch slices of FULL-batch tensors (S * N items) - the S fusion steps of a sequence, whose forward
    passes depend on each other through the detached flow but whose backward passes do not: every operator allocates its
    full-batch output and records its closure ONCE (first step), later steps only fill their slice; one backward over the
    full batch then replaces S backward passes (S x fewer launches, S x deeper weight-gradient reductions; same sums)."""

    def __init__(self):
        self.fns = []
        self.k = None

    def begin_steps(self, S, N):
        self.S, self.N, self.k, self.pos, self.nodes = S, N, 0, 0, []

    def step(self, k):
        self.k, self.pos = k, 0

    def end_steps(self):
        self.k = None

    def _sl(self):
        return slice(self.k * self.N, (self.k + 1) * self.N)

    def _run(self, ins, fwd, mk_bw, needs=True):
        """fwd(list of input tensors, out tensor or None) -> result tensor; mk_bw(out Var, y) -> closure."""
        if self.k is None:
            y = fwd([v.v for v in ins], None)
            out = Var(y, needs)
            self.fns.append(mk_bw(out, y))
            return out
        sl = self._sl()
        xs = [v.v[sl] for v in ins]
        if self.k == 0:
            r = fwd(xs, None)                       # (the first slice's result fixes the output shape)
            y = torch.empty((self.S * self.N,) + tuple(r.shape[1:]), dtype=torch.float32, device=r.device)
            ops.copy_into(r, y[sl])
            out = Var(y, needs)
            self.fns.append(mk_bw(out, y))
            self.nodes.append(out)
            self.pos += 1
            return out
        out = self.nodes[self.pos]
        self.pos += 1
        r = fwd(xs, out.v[sl])
        if r.data_ptr() != out.v[sl].data_ptr():
            ops.copy_into(r, out.v[sl])
        return out

    def backward(self, keep=False):
        """keep=True leaves the closures (and the activations they hold) alive: the caller ran this on another stream than
        the one the activations were allocated on and drops the tape once that stream has been joined."""
        for fn in reversed(self.fns):
            fn()
        if not keep:
            self.fns = []

    @staticmethod
    def _pacc(p, g):
        if p is None or g is None:
            return
        g = g.reshape(p.shape)
        if p.grad is None:
            p.grad = g.contiguous()
        else:
            p.grad.add_(g)

    # ---- operators --------------------------------------------------------------------------------------------------
    def conv(self, x, conv, pk, act=ops.ACT_NONE):
        """nn.Conv2d (+ fused relu / sigmoid) with the pack `pk` of its current weights."""
        KH, KW = conv.kernel_size
        st, pad = conv.stride[0], tuple(conv.padding)

        def mk_bw(out, y):
            def bw():
                if out.g is None:
                    return
                g = B.act_backward(out.g, y, act) if act != ops.ACT_NONE else out.g
                with _param_grad_stream(g, x.v):
                    dw, db = B.conv_wgrad(x.v, g, KH, KW, stride=st, padding=pad, bias=conv.bias is not None)
                    self._pacc(conv.weight, dw)
                    self._pacc(conv.bias, db)
                if x.needs:
                    x.acc(B.conv_dgrad(g, conv.weight, tuple(x.v.shape[2:]), stride=st, padding=pad, deps=(conv.weight,)))
            return bw
        return self._run([x], lambda xs, o: ops.conv2d(pk, xs[0], act=act, out=o), mk_bw)

    def zero_conv(self, x, zc, pk):
        """ZeroConv2d (modules.py:94-96): out = conv(x) * exp(3 scale); the pack has the factor folded into the weights."""
        conv = zc.conv
        KH, KW = conv.kernel_size
        pad = tuple(conv.padding)

        def mk_bw(out, y):
            def bw():
                if out.g is None:
                    return
                g = out.g
                e = torch.exp(zc.scale.detach().float() * 3).reshape(-1)
                with _param_grad_stream(g, x.v, y, e):
                    dw, db = B.conv_wgrad(x.v, g, KH, KW, padding=pad)          # gradients w.r.t. the FOLDED weights
                    self._pacc(conv.weight, dw * e.view(-1, 1, 1, 1))
                    self._pacc(conv.bias, db * e)
                    # d/d scale_c = 3 sum_{b,y,x} g_c out_c: the diagonal of the 1x1 "weight gradient" of out against g
                    dd, _ = B.conv_wgrad(y, g, 1, 1, bias=False)
                    self._pacc(zc.scale, 3.0 * torch.diagonal(dd.reshape(dd.shape[0], dd.shape[1])))
                if x.needs:
                    x.acc(B.conv_dgrad(g, lambda: conv.weight.detach().float() * e.view(-1, 1, 1, 1), tuple(x.v.shape[2:]),
                                       padding=pad, deps=(conv.weight, zc.scale)))
            return bw
        return self._run([x], lambda xs, o: ops.conv2d(pk, xs[0], out=o), mk_bw)

    def cat(self, parts):
        """torch.cat(dim=1) into one buffer (HIP copies, as the fp32 inference path lays its concatenations out)."""
        cs = [p.v.shape[1] for p in parts]

        def fwd(xs, o):
            Bn, _, H, W = xs[0].shape
            buf = o if o is not None else torch.empty((Bn, sum(cs), H, W), dtype=torch.float32, device=xs[0].device)
            c0 = 0
            for x, c in zip(xs, cs):
                ops.copy_into(x, buf[:, c0:c0 + c])
                c0 += c
            return buf

        def mk_bw(out, y):
            def bw():
                if out.g is None:
                    return
                Bn, _, H, W = y.shape
                c0 = 0
                for p, c in zip(parts, cs):
                    if p.needs:
                        sl = out.g[:, c0:c0 + c]
                        # a part that already has a gradient is added to; otherwise it takes a private copy (later add_
                        # calls on the part write in place, and the slice shares this buffer's storage)
                        if p.g is None:
                            own = torch.empty((Bn, c, H, W), dtype=torch.float32, device=y.device)
                            ops.copy_into(sl, own)
                            p.g = own
                        else:
                            B.add_(p.g, sl)
                    c0 += c
            return bw
        return self._run(parts, fwd, mk_bw, needs=any(p.needs for p in parts))

    def batch_slices(self, x, n):
        """split along the batch into len(x) // n parts (FlowEncoder's list call, AccFlow_.py:60-67)."""
        k = x.v.shape[0] // n
        outs = [Var(x.v[i * n:(i + 1) * n]) for i in range(k)]

        def bw():
            if not x.needs or all(o.g is None for o in outs):
                return
            g = torch.zeros_like(x.v)
            for i, o in enumerate(outs):
                if o.g is not None:
                    ops.copy_into(o.g, g[i * n:(i + 1) * n])
            x.acc(g)
        self.fns.append(bw)
        return outs

    def split_offsets_mask(self, om):
        """AccFlow_.py:102-103: split [18, 9] of the 27 channels, sigmoid on the 9 mask channels (out of place: the
        ZeroConv2d backward needs its own output)."""
        def make():
            off = Var(om.v[:, :18])
            mv = torch.empty((om.v.shape[0], om.v.shape[1] - 18) + tuple(om.v.shape[2:]), dtype=torch.float32, device=om.v.device)
            msk = Var(mv)

            def bw():
                if off.g is None and msk.g is None:
                    return
                d = torch.zeros_like(om.v)
                if off.g is not None:
                    ops.copy_into(off.g, d[:, :18])
                if msk.g is not None:
                    B.act_backward(msk.g, msk.v, ops.ACT_SIGMOID, out=d[:, 18:])
                om.acc(d)
            self.fns.append(bw)
            return off, msk
        if self.k is None:
            off, msk = make()
            sl = slice(None)
        else:
            if self.k == 0:
                self.nodes.append(make())
            off, msk = self.nodes[self.pos]
            self.pos += 1
            sl = self._sl()
        ops.copy_into(om.v[sl][:, 18:], msk.v[sl])
        ops.activation_(msk.v[sl], ops.ACT_SIGMOID)
        return off, msk

    def deform_conv(self, x, off, msk, dconv, pk):
        def mk_bw(out, y):
            def bw():
                if out.g is None:
                    return
                dx, doff, dm, dw, db = B.deform_conv_backward(x.v, off.v, msk.v, dconv.weight, out.g, need_dx=x.needs,
                                                              deps=(dconv.weight,))
                with _param_grad_stream(dw, db):
                    self._pacc(dconv.weight, dw)
                    self._pacc(dconv.bias, db)
                if dx is not None:
                    x.acc(dx)
                off.acc(doff)
                msk.acc(dm)
            return bw
        return self._run([x, off, msk], lambda xs, o: ops.conv2d(pk, xs[0], offset=xs[1], dmask=xs[2], out=o), mk_bw)

    def add_relu(self, a, b):
        """relu(a + b): the residual join of extractor.py:44-47."""
        y = torch.empty_like(a.v)
        ops.copy_into(a.v, y)
        B.add_(y, b.v)
        ops.activation_(y, ops.ACT_RELU)
        out = Var(y)

        def bw():
            if out.g is None:
                return
            g = B.act_backward(out.g, y, ops.ACT_RELU)
            a.acc(g)
            if b.needs:
                if b.g is None:          # a and b must not share one gradient tensor (later add_ calls are in place)
                    own = torch.empty_like(g)
                    ops.copy_into(g, own)
                    b.g = own
                else:
                    B.add_(b.g, g)
        self.fns.append(bw)
        return out

    def blend(self, f1, f2, m):
        def mk_bw(out, y):
            def bw():
                if out.g is None:
                    return
                d1, d2, dm = B.blend_backward(out.g, f1.v, f2.v, m.v)
                f1.acc(d1)
                f2.acc(d2)
                m.acc(dm)
            return bw
        return self._run([f1, f2, m], lambda xs, o: ops.blend(xs[0], xs[1], xs[2]), mk_bw)

    def convex_upsample(self, flow, mask):
        def mk_bw(out, y):
            def bw():
                if out.g is None:
                    return
                dflow, dmask = B.convex_upsample_backward(out.g, flow.v, mask.v)
                flow.acc(dflow)
                mask.acc(dmask)
            return bw
        return self._run([flow, mask], lambda xs, o: ops.convex_upsample(xs[0], xs[1], out=o), mk_bw)


# ---- the modules of the fusion step on the tape (same packs / caches as their inference forwards) ---------------------------
def flow_encoder_fw(t, m, flows):
    """FlowEncoder.forward on a list (AccFlow_.py:56-67); the flows carry no gradient."""
    n = flows[0].shape[0]
    x = Var(torch.cat([f.float() for f in flows], dim=0).contiguous(), needs=False)
    pk = m._packs
    x = t.conv(x, m.conv1, pk.conv("1", m.conv1), act=ops.ACT_RELU)
    x = t.conv(x, m.conv2, pk.conv("2", m.conv2), act=ops.ACT_RELU)
    x = t.conv(x, m.conv3, pk.conv("3", m.conv3))
    return t.batch_slices(x, n)


def context_fw(t, m, image):
    """BasicEncoder(norm_fn='none') (extractor.py:117-175 with identity norms) of ONE frame batch."""
    if m.norm_fn != "none" or m.dropout is not None:
        raise NotImplementedError("training slice: the context encoder of AccFlow (norm_fn='none', no dropout)")
    pk = m._packs
    x = t.conv(Var(image.float().contiguous(), needs=False), m.conv1, pk.conv("stem", m.conv1), act=ops.ACT_RELU)
    for li in (1, 2, 3):
        for bi, blk in enumerate(getattr(m, "layer%d" % li)):
            tag = "l%d.%d" % (li, bi)
            y = t.conv(x, blk.conv1, pk.conv(tag + ".c1", blk.conv1), act=ops.ACT_RELU)
            y = t.conv(y, blk.conv2, pk.conv(tag + ".c2", blk.conv2), act=ops.ACT_RELU)
            if blk.downsample is not None:
                x = t.conv(x, blk.downsample[0], pk.conv(tag + ".ds", blk.downsample[0]))
            x = t.add_relu(x, y)
    return t.conv(x, m.conv2, pk.conv("head", m.conv2))


def accplus_fw(t, m, df, f, o, c):
    """AccPlus.forward (AccFlow_.py:97-109)."""
    pk = m._packs
    x = t.cat([df, f, o])
    x = t.conv(t.conv(x, m.conv1[0], pk.conv("1a", m.conv1[0]), act=ops.ACT_RELU), m.conv1[2], pk.conv("1b", m.conv1[2]))
    x = t.cat([x, c])
    x = t.conv(x, m.conv2[0], pk.conv("2a", m.conv2[0]), act=ops.ACT_RELU)
    x = t.conv(x, m.conv2[2], pk.conv("2b", m.conv2[2]), act=ops.ACT_RELU)
    zc = m.conv2[4]
    om = t.zero_conv(x, zc, pk.conv("2z", zc.conv, scale=zc.out_scale, scale_dep=zc.scale))
    off, msk = t.split_offsets_mask(om)
    f_ = t.deform_conv(f, off, msk, m.dconv, pk.conv("dc", m.dconv_as_conv(), tap_major=True))
    x = t.cat([f_, df, o])
    x = t.conv(t.conv(x, m.conv3[0], pk.conv("3a", m.conv3[0]), act=ops.ACT_RELU), m.conv3[2], pk.conv("3b", m.conv3[2]))
    x = t.cat([x, c, f_, df])
    x = t.conv(x, m.conv4[0], pk.conv("4a", m.conv4[0]), act=ops.ACT_RELU)
    x = t.conv(x, m.conv4[2], pk.conv("4b", m.conv4[2]), act=ops.ACT_RELU)
    return t.conv(x, m.conv4[4], pk.conv("4c", m.conv4[4]))


def blending_mask_fw(t, m, emap):
    """The mask of Blending.forward (AccFlow_.py:119-121); the error map is detached (AccFlow_.py:198)."""
    pk = m._packs
    x = t.conv(Var(emap.float().contiguous(), needs=False), m.mask[0], pk.conv("0", m.mask[0]), act=ops.ACT_RELU)
    return t.conv(x, m.mask[2], pk.conv("2", m.mask[2]), act=ops.ACT_SIGMOID)


def blending_fw(t, m, f1, f2, emap):
    """Blending.forward (AccFlow_.py:118-124)."""
    return t.blend(f1, f2, blending_mask_fw(t, m, emap))


def flow_decoder_fw(t, m, x):
    """FlowDecoder.forward (AccFlow_.py:38-45)."""
    pk = m._packs
    fl = t.conv(t.conv(x, m.flow[0], pk.conv("f0", m.flow[0]), act=ops.ACT_RELU), m.flow[2], pk.conv("f2", m.flow[2]))
    mk = t.conv(t.conv(x, m.mask[0], pk.conv("m0", m.mask[0]), act=ops.ACT_RELU), m.mask[2], pk.conv("m2", m.mask[2]))
    return fl, t.convex_upsample(fl, mk)


def trainable_parameters(model):
    """What train_acc.py:163-173 hands the optimizer: everything but the frozen estimator."""
    return [p for n, p in model.named_parameters() if not n.startswith("ofe.")]


def fusion_step_fw(t, model, I1, I2, In, F2n, flows=None, ctx=None, hoisted=None, mode=None):
    """AccFlow.iter (AccFlow_.py:177-201) with the gradient-carrying part on the tape.  -> (flow_small Var, flow_up Var).
    flows = (dflow, flow_ini[, F2n]) at 1/8 resolution when the caller estimated them already (forward_backward: all pairs
    of the sequence in one batched estimator call, as the inference path does; per-sample identical to the reference's
    per-step calls - InstanceNorm and eval-mode BatchNorm do not mix samples).  ctx = (c1 Var, c2, cn) when the caller
    encoded the context features already (c1 on a tape of its own, see forward_backward); hoisted = (f_ini Var, df Var,
    blending-mask Var) likewise.
    mode: arithmetic of the heads' forward.  None = the guarded training mode (TRAIN_CONV_MODE) when the call runs inside a
    guard scope somebody reads (forward_backward, GraphedForwardBackward), else bf16x6 - which has no range condition: a
    bare call never computes on an fp16 split whose range reports nobody looks at."""
    from .networks.AccFlow_ import downflow8, getOcc
    if mode is None:
        mode = TRAIN_CONV_MODE if (TRAIN_CONV_MODE != "f16x3" or ops.inside_guard()) else "bf16x6"
    with torch.no_grad():
        if flows is not None:
            dflow, flow_ini = flows[0], flows[1]
            F2n = flows[2] if F2n is None else F2n
        elif F2n is None:
            dflow, flow_ini, F2n = downflow8(model.ofe(torch.cat([I1, I1, I2]), torch.cat([I2, In, In]))).chunk(3)
        else:
            dflow, flow_ini = downflow8(model.ofe(torch.cat([I1, I1]), torch.cat([I2, In]))).chunk(2)
        if ctx is None:
            c2, cn = model.context([I2, In])      # reach the loss through detached maps only (AccFlow_.py:195,198)
    with ops.conv_mode(mode):
        if hoisted is None:
            f_ini, df, f = flow_encoder_fw(t, model.flow_encoder, [flow_ini, dflow, F2n])
        else:
            f_ini, df = hoisted[0], hoisted[1]
            f = flow_encoder_fw(t, model.flow_encoder, [F2n])[0]
        if ctx is None:
            c1 = context_fw(t, model.context, I1)
        else:
            c1, c2, cn = ctx
        o = Var(getOcc(dflow.contiguous(), c1.v, c2), needs=False)
        f_acc = accplus_fw(t, model.accplus, df, f, o, c1)
        if hoisted is None:
            f_fuse = blending_fw(t, model.blending, f_ini, f_acc, getOcc(flow_ini.contiguous(), c1.v, cn, binary=False))
        else:
            f_fuse = t.blend(f_ini, f_acc, hoisted[2])
        return flow_decoder_fw(t, model.flow_decoder, f_fuse)


def forward_backward(model, images, flow_gts, sync_loss=True, small=None):
    """_forward_backward under the f16x3 range guard (TRAIN_CONV_MODE): one device flag for the whole step, read behind the
    loss's host synchronisation; a tripped step is undone (param.grad restored) and redone in bf16x6.  Inside an enclosing
    guard scope (GraphedForwardBackward owns the flag of its replay) the caller reads the flag.  With sync_loss=False and NO
    enclosing scope nobody would read it: that call runs in bf16x6, which has no range condition."""
    if TRAIN_CONV_MODE != "f16x3" or ops.inside_guard():
        return _forward_backward(model, images, flow_gts, sync_loss, TRAIN_CONV_MODE, small)
    if not sync_loss:
        with ops.conv_mode("bf16x6"):
            return _forward_backward(model, images, flow_gts, False, "bf16x6", small)
    params = trainable_parameters(model)
    saved = [None if p.grad is None else p.grad.detach().clone() for p in params]
    flag = torch.zeros(1, dtype=torch.int32, device=images[0].device)
    with ops.conv_mode("f16x3"), ops.guard_scope(flag):
        loss, outs = _forward_backward(model, images, flow_gts, True, "f16x3", small)
    if not int(flag.item()):
        return loss, outs
    ops.note_guard_trip("train.forward_backward")
    for p, g in zip(params, saved):
        p.grad = g
    with ops.conv_mode("bf16x6"):
        return _forward_backward(model, images, flow_gts, True, "bf16x6", small)


def _forward_backward(model, images, flow_gts, sync_loss=True, TRAIN_CONV_MODE=None, small=None):
    """The loss of train_acc.py:223-224 on one sequence and its gradients: images [I_0 .. I_n], flow_gts [gt of F(2->0) ..
    F(n->0)] (full resolution).  Adds into `param.grad` of the trainable parameters.  -> (loss, predictions).

    What does not depend on the accumulated flow runs once per sequence instead of once per fusion step, in the inference
    path's batching: the frozen estimator for all pairs (AccFlow.estimate_small); the context encoder - the detached c2 / cn
    features of all frames in one inference call, and the gradient-carrying c1 features of frames 2..n as ONE taped batch;
    FlowEncoder of every step's flow_ini / dflow and the blending masks (their inputs are detached: AccFlow_.py:183,198)
    likewise.  That shared tape's backward runs once, after every step's backward has delivered its share of the
    gradients (the steps are tied only through the parameters, AccFlow_.py:171-172: the same sums in another order)."""
    if len(flow_gts) != len(images) - 2:
        raise ValueError("length not match!")          # loss.py:32
    if TRAIN_CONV_MODE is None:
        TRAIN_CONV_MODE = globals()["TRAIN_CONV_MODE"]
    images = list(images)
    N = images[0].shape[0]
    steps = list(range(2, len(images)))
    pairs = model.pair_schedule(len(images))
    with torch.no_grad():
        if small is None:
            small = model.estimate_small(images, pairs)
            by_pair = {p: small[k * N:(k + 1) * N].contiguous() for k, p in enumerate(pairs)}
        else:
            # the frozen estimator's 1/8-resolution flows handed in ({(i, j): (N, 2, h, w)}: tests pin the gradient-carrying
            # heads on the reference's own estimator outputs, tests/golden/make_grad_golden.py)
            by_pair = {p: small[p].float().contiguous() for p in pairs}
        ctx_all = model.context([im.float().contiguous() for im in images])
    from .networks.AccFlow_ import getOcc
    tc, S = Tape(), len(steps)
    with ops.conv_mode(TRAIN_CONV_MODE):
        c1_cat = context_fw(tc, model.context, torch.cat([images[i] for i in steps], dim=0))
        c1_all = None if BATCHED_BACKWARD else tc.batch_slices(c1_cat, N)
        flow_ini_all = torch.cat([by_pair[(i, 0)] for i in steps], dim=0)
        fe = flow_encoder_fw(tc, model.flow_encoder, [flow_ini_all, torch.cat([by_pair[(i, i - 1)] for i in steps], dim=0)])
        f_ini_all, df_all = (None, None) if BATCHED_BACKWARD else (tc.batch_slices(fe[0], N), tc.batch_slices(fe[1], N))
        emap = getOcc(flow_ini_all.contiguous(), c1_cat.v, ctx_all[0].repeat(S, 1, 1, 1), binary=False)
        m_all_full = blending_mask_fw(tc, model.blending, emap)
        m_all = None if BATCHED_BACKWARD else tc.batch_slices(m_all_full, N)
    dev = images[0].device
    if BATCHED_BACKWARD:
        # The S fusion steps run in the tape's STEP MODE on full-batch tensors (S * N items): forward step by step (step k
        # reads step k-1's detached 1/8-resolution flow), ONE backward over all steps at the end.
        h, w = by_pair[(1, 0)].shape[2:]
        dflow_all = torch.cat([by_pair[(i, i - 1)] for i in steps], dim=0).contiguous()
        c2_all = torch.cat([ctx_all[i - 1] for i in steps], dim=0)
        with ops.conv_mode(TRAIN_CONV_MODE):
            o_all = Var(getOcc(dflow_all, c1_cat.v, c2_all), needs=False)
            m_full = m_all_full
            F2n = Var(torch.empty((S * N, 2, h, w), dtype=torch.float32, device=dev), needs=False)
            pkf, fe_m = model.flow_encoder._packs, model.flow_encoder
            tc.begin_steps(S, N)
            prev = by_pair[(1, 0)]
            for k in range(S):
                tc.step(k)
                ops.copy_into(prev.contiguous(), F2n.v[k * N:(k + 1) * N])
                f = tc.conv(F2n, fe_m.conv1, pkf.conv("1", fe_m.conv1), act=ops.ACT_RELU)
                f = tc.conv(f, fe_m.conv2, pkf.conv("2", fe_m.conv2), act=ops.ACT_RELU)
                f = tc.conv(f, fe_m.conv3, pkf.conv("3", fe_m.conv3))
                f_acc = accplus_fw(tc, model.accplus, fe[1], f, o_all, c1_cat)
                f_fuse = tc.blend(fe[0], f_acc, m_full)
                small, up = flow_decoder_fw(tc, model.flow_decoder, f_fuse)
                prev = small.v[k * N:(k + 1) * N]
            tc.end_steps()
            gt_all = torch.cat([g.float() for g in flow_gts], dim=0).contiguous()
            loss = (up.v - gt_all).abs().mean() * S          # sum_k mean |F_k - gt_k| (equal sizes), loss.py:34-36
            up.g = B.l1_grad(up.v, gt_all, float(S) / up.v.numel())
            tc.backward()
        join_param_grads(dev)
        return (float(loss) if sync_loss else loss), list(up.v.split(N, dim=0))
    main = torch.cuda.current_stream(dev)
    side = _backward_stream(dev) if OVERLAP_BACKWARD else None
    flow, loss, outs, tapes = None, 0.0, [], []
    for k, i in enumerate(steps):
        t = Tape()
        small_k, up = fusion_step_fw(t, model, images[i], images[i - 1], images[0], flow,
                                     flows=(by_pair[(i, i - 1)], by_pair[(i, 0)], by_pair[(1, 0)]),
                                     ctx=(c1_all[k], ctx_all[i - 1], ctx_all[0]), hoisted=(f_ini_all[k], df_all[k], m_all[k]),
                                     mode=TRAIN_CONV_MODE)
        gt = flow_gts[k].float().contiguous()
        loss = loss + (up.v - gt).abs().mean()           # the value of loss.py:34-36 (its gradient: l1_grad below)
        if side is not None:
            side.wait_stream(main)                       # step k's forward (and everything before it) is complete for `side`
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext(), ops.conv_mode(TRAIN_CONV_MODE):
            up.g = B.l1_grad(up.v, gt, 1.0 / up.v.numel())
            t.backward(keep=side is not None)
        tapes.append((t, gt))                            # activations allocated on `main`, read on `side`: alive until the join
        flow = small_k.v                                 # detached between steps (AccFlow_.py:171-172)
        outs.append(up.v)
    if side is not None:
        main.wait_stream(side)
    with ops.conv_mode(TRAIN_CONV_MODE):
        tc.backward()
    join_param_grads(dev)
    if side is not None:
        # (the tapes die here, after the join was enqueued on `main`: their blocks return to main's pool, whose next user is
        # ordered behind the join)
        tapes.clear()
    return (float(loss) if sync_loss else loss), outs


class GraphedForwardBackward:
    """forward_backward captured ONCE in a HIP graph (torch.cuda.graph) for fixed shapes and replayed: no Python between the
    ~2 600 launches of a step.  Measured 51.5 vs 53.2 ms per step (tools/train_bench.py): the step is bound by kernel time
    on the GPU (51 ms of a step have at least one kernel running), the host's launch rate is a close second.  The front
    end's default since round 5 (train_acc.py: ACCFLOW_TRAIN_GRAPH=0 for the eager step).
    Inputs are copied into static buffers; parameter gradients land in static tensors that are re-attached to `param.grad`
    after every replay; the weight packs of the trainable modules are rebuilt INSIDE the graph (their pack kernels are part
    of it), so every replay sees the parameters the optimizer just updated; the frozen estimator's packs stay cached.
    Everything runs inside one ops.guard_scope whose flag is read after the replay: the frozen estimator, the context
    encoder AND the heads' forward run on the guarded fp16 split (TRAIN_CONV_MODE); if a value left its range, the step is
    redone eagerly, ONCE, in bf16x6 (two passes in all, not three).  The training loader serves fixed shapes (drop_last,
    fixed crop)."""

    def __init__(self, model, images, flow_gts, warmup=2):
        self.model = model
        self.params = trainable_parameters(model)
        self.images = [im.detach().clone() for im in images]
        self.gts = [g.detach().float().clone() for g in flow_gts]
        dev = self.images[0].device
        saved = [p.grad for p in self.params]
        for _ in range(warmup):                      # builds every cache (estimator packs, workspaces) outside the graph
            for p in self.params:
                p.grad = None
            forward_backward(model, self.images, self.gts)
        for m in (model.flow_encoder, model.context, model.accplus, model.blending, model.flow_decoder):
            m._packs.clear()                         # trainable packs: rebuilt by kernels that belong to the graph
        B._DGRAD_CACHE.clear()
        for p in self.params:
            p.grad = None
        self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            with ops.guard_scope(self.flag):
                self.loss_t, self.outs = forward_backward(model, self.images, self.gts, sync_loss=False)
        self.grads = [p.grad for p in self.params]
        # The captured launches hold raw pointers into caches that live OUTSIDE the graph's private pool: the split-K scratch
        # buffers (ops._ksplit_ws frees and re-allocates one when a later eager call on the same stream asks for more, e.g. a
        # full-resolution validation) - keep them alive for as long as the graph is.
        self._pinned = list(getattr(ops._tls, "ksplit_ws", {}).values())
        # Capture only RECORDS: the pack / transposed-pack entries of the trainable modules now carry valid signatures but
        # point at memory no kernel has written.  One replay fills them, so an eager forward before the first step is safe.
        self.graph.replay()
        torch.cuda.synchronize(dev)
        for p, g in zip(self.params, saved):
            p.grad = g

    def shapes_match(self, images, flow_gts):
        return (len(images) == len(self.images) and len(flow_gts) == len(self.gts)
                and all(a.shape == b.shape for a, b in zip(images, self.images))
                and all(a.shape == b.shape for a, b in zip(flow_gts, self.gts)))

    def __call__(self, images, flow_gts):
        for s, x in zip(self.images, images):
            s.copy_(x)
        for s, x in zip(self.gts, flow_gts):
            s.copy_(x)
        self.flag.zero_()
        self.graph.replay()
        for p, g in zip(self.params, self.grads):
            p.grad = g
        loss = float(self.loss_t)                    # (synchronises with the replay)
        if int(self.flag.item()):                    # rare: redo eagerly, straight in bf16x6 (no second f16x3 attempt)
            for p in self.params:
                p.grad = None
            ops.note_guard_trip("train.GraphedForwardBackward")
            with ops.conv_mode("bf16x6"):
                return _forward_backward(self.model, images, flow_gts, True, "bf16x6")
        return loss, self.outs


def allreduce_grads(params, group=None):
    """Data-parallel training, one process per GPU: average the gradients over the ranks in ONE collective (the ~6.8 M
    trainable parameters = 27 MB as a single fp32 bucket - a ring all-reduce over xGMI is per-link bound, so one large
    message beats per-tensor ones).  The reference's nn.DataParallel (train_acc.py:166) scatters the batch over its GPUs
    and sums the replicas' gradients of ONE loss over the whole batch (loss.py:34-36 takes the mean over it); the mean
    of the ranks' equal-sized per-rank means is the same number.  A parameter without a gradient on this rank enters as
    zeros so that every rank issues the same collective."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return
    world = dist.get_world_size(group)
    if world == 1:
        return
    params = list(params)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in params])
    dist.all_reduce(flat, group=group)
    flat /= world
    o = 0
    for p in params:
        n = p.numel()
        g = flat[o:o + n].view(p.shape)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        o += n


def train_step(model, optimizer, images, flow_gts, clip=1.0, scheduler=None, group=None, graphed=None):
    """optimizer.zero_grad / forward / backward / clip / step of train_acc.py:210-234 (no GradScaler: nothing here
    computes in fp16); with torch.distributed initialised the gradients are averaged over the ranks first.
    graphed: a GraphedForwardBackward for these shapes (replayed instead of the eager forward / backward)."""
    optimizer.zero_grad(set_to_none=True)
    loss, outs = graphed(images, flow_gts) if graphed is not None else forward_backward(model, images, flow_gts)
    allreduce_grads(trainable_parameters(model), group)
    torch.nn.utils.clip_grad_norm_(trainable_parameters(model), clip)
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return loss, outs
