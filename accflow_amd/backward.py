"""Backward operators of the fusion heads on the HIP library.

The reference trains AccFlow with torch autograd (train_acc.py:222-233): the estimator is frozen (train_acc.py:164) and the
flows / occlusion / error maps are detached (AccFlow_.py:172,182,195,198), so gradients flow through the convolution
stacks of FlowEncoder / AccPlus / Blending / FlowDecoder / the context encoder, the modulated deformable convolution, the
blend and the convex upsampling only.  These wrappers are the gradients of exactly those operators, fp32:

  conv_dgrad      input gradient  = the FORWARD convolution kernel with the transposed, flipped weights (stride 2: over the
                  zero-inserted output gradient, accflow_dilate_f32)
  conv_wgrad      weight / bias gradient (accflow_conv_wgrad_f32)
  act_backward    relu / sigmoid / tanh from the activation's output
  deform_conv_backward, blend_backward, convex_upsample_backward, l1_grad, add_

Parity is checked per operator against torch autograd of the same op in float64 (tests/test_backward.py) and end to end
against the reference's own autograd through committed gradient fixtures (tests/golden/make_grad_golden.py)."""
import torch

from . import _lib, ops, profiler
from .ops import _check, _p, _plane4, _stream


def _chw(t):
    return t.shape[1] * t.shape[2] * t.shape[3]


def act_backward(dy, y, act, out=None):
    """dy * act'(y) from the activation's output y; operands may be channel slices ((B, C, H, W) with a batch stride)."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(tuple(dy.shape), dtype=torch.float32, device=dy.device)
    if tuple(dy.shape) != tuple(y.shape) or tuple(out.shape) != tuple(y.shape):
        raise RuntimeError("act_backward: shapes differ")
    _check(lib.accflow_act_backward_f32(_p(dy), _plane4(dy, "dy"), _p(y), _plane4(y, "y"), _p(out), _plane4(out, "dx"),
                                        dy.shape[0], _chw(dy), int(act), _stream()), "accflow_act_backward_f32")
    return out


def add_(dst, src):
    """dst += src (both (B, C, H, W), batch strides allowed)."""
    lib = _lib.load()
    if tuple(dst.shape) != tuple(src.shape):
        raise RuntimeError("add_: shapes differ")
    _check(lib.accflow_add_f32(_p(dst), _plane4(dst, "dst"), _p(src), _plane4(src, "src"), dst.shape[0], _chw(dst), _stream()),
           "accflow_add_f32")
    return dst


def l1_grad(pred, gt, scale):
    """scale * sign(pred - gt): gradient of scale * sum |pred - gt| (loss.py:34-36 with scale = 1 / numel)."""
    lib = _lib.load()
    pred, gt = ops._dense(pred, "pred"), ops._dense(gt, "gt")
    out = torch.empty_like(pred)
    _check(lib.accflow_l1_grad_f32(_p(pred), _p(gt), _p(out), pred.numel(), float(scale), _stream()), "accflow_l1_grad_f32")
    return out


def blend_backward(dy, f1, f2, m):
    """Blending (AccFlow_.py:122-124) out = f1 m + (1 - m) f2 -> (df1, df2, dm)."""
    lib = _lib.load()
    dy, f1, f2, m = (ops._dense(t, n) for t, n in ((dy, "dy"), (f1, "f1"), (f2, "f2"), (m, "m")))
    B, C, H, W = f1.shape
    df1, df2 = torch.empty_like(f1), torch.empty_like(f1)
    dm = torch.empty((B, 1, H, W), dtype=torch.float32, device=f1.device)
    _check(lib.accflow_blend_backward_f32(_p(dy), _p(f1), _p(f2), _p(m), _p(df1), _p(df2), _p(dm), B, C, H * W, _stream()),
           "accflow_blend_backward_f32")
    return df1, df2, dm


def convex_upsample_backward(dup, flow, mask):
    """raft.py:81-92 backward: dup (B, 2, 8 H8, 8 W8) -> (dflow (B, 2, H8, W8), dmask (B, 576, H8, W8))."""
    lib = _lib.load()
    dup, flow, mask = ops._dense(dup, "dup"), ops._dense(flow, "flow"), ops._dense(mask, "mask")
    B, _, H8, W8 = flow.shape
    if tuple(dup.shape) != (B, 2, 8 * H8, 8 * W8) or tuple(mask.shape) != (B, 576, H8, W8):
        raise RuntimeError("convex_upsample_backward: dup (B,2,8H8,8W8), mask (B,576,H8,W8)")
    dflow, dmask = torch.empty_like(flow), torch.empty_like(mask)
    _check(lib.accflow_convex_upsample_backward_f32(_p(dup), _p(flow), _p(mask), _p(dflow), _p(dmask), B, H8, W8, _stream()),
           "accflow_convex_upsample_backward_f32")
    return dflow, dmask


def conv_wgrad(x, dy, KH, KW, stride=1, padding=(0, 0), bias=True):
    """-> (dw (Cout, Cin, KH, KW), db (Cout) or None) of y = conv2d(x, w, b, stride, padding) given dy."""
    lib = _lib.load()
    xbs, dbs = _plane4(x, "x"), _plane4(dy, "dy")
    B, Cin, H, W = x.shape
    Cout = dy.shape[1]
    pH, pW = (padding, padding) if isinstance(padding, int) else padding
    OH, OW = (H + 2 * pH - KH) // stride + 1, (W + 2 * pW - KW) // stride + 1
    if tuple(dy.shape) != (B, Cout, OH, OW):
        raise RuntimeError("conv_wgrad: dy is %s, expected %s" % (tuple(dy.shape), (B, Cout, OH, OW)))
    dw = torch.empty((Cout, Cin, KH, KW), dtype=torch.float32, device=x.device)
    db = torch.empty((Cout,), dtype=torch.float32, device=x.device) if bias else None
    tm = profiler.ACTIVE
    t0 = tm.begin() if tm is not None and tm.wants("conv_wgrad") else None
    _check(lib.accflow_conv_wgrad_f32(_p(x), xbs, _p(dy), dbs, _p(dw), _p(db), B, Cin, Cout, H, W, KH, KW, int(stride), int(pH),
                                      int(pW), _stream()), "accflow_conv_wgrad_f32")
    if t0 is not None:   # algorithmic flop of dW[co][ci, tap] = sum_p dY[co][p] X[ci, tap][p]
        tm.end("conv_wgrad", t0, 2.0 * Cout * Cin * KH * KW * B * OH * OW, "wgrad Cin%d Cout%d k%dx%d s%d B%d %dx%d" % (
            Cin, Cout, KH, KW, stride, B, H, W))
    return dw, db


GRAD_CONV_MODE = "bf16x6"   # the input-gradient convolutions: fp32-equivalent products with no range condition

_DGRAD_CACHE = {}   # (tag, ids of the parameters a weight derives from) -> [their (data_ptr, version), value]: the five fusion
#                     steps of a training step differentiate the same weights; rebuilt when a version moves


def _cached(deps, tag, build):
    if deps is None:
        return build()
    key = (tag, tuple(id(d) for d in deps))
    sig = tuple((d.data_ptr(), d._version, str(d.device)) for d in deps)
    hit = _DGRAD_CACHE.get(key)
    if hit is None or hit[0] != sig:
        hit = _DGRAD_CACHE[key] = [sig, build()]
    return hit[1]


def conv_dgrad(dy, weight, in_hw, stride=1, padding=(0, 0), out=None, deps=None):
    """Input gradient of y = conv2d(x, weight, stride, padding): the forward convolution kernel over dy (zero-inserted
    for stride > 1) with weight^T flipped in both axes and padding K - 1 - p.  in_hw = (H, W) of x.  deps: the
    parameters `weight` is, or is derived from - the transposed pack is then kept until one of them changes; a callable
    `weight` (a derived tensor: ZeroConv2d's folded weights, the deformable convolution's column form) is evaluated only
    when they did."""
    lib = _lib.load()
    pH, pW = (padding, padding) if isinstance(padding, int) else padding
    H, W = in_hw
    B = dy.shape[0]
    if callable(weight):
        weight = _cached(deps, "w", weight)
    Cout, Cin, KH, KW = weight.shape
    pad = (KH - 1 - pH, KW - 1 - pW)
    def make_pack():
        # the input-gradient convolution's weights W^T (flipped): packed straight from the forward weight where the fused pack
        # entry point can (more than 4 input channels); else through a transposed copy (parameter-sized plumbing)
        if ops.PACK_FUSED and Cin > 4:
            return ops.PackedConv(weight.detach().float(), None, stride=1, padding=pad, transpose_flip=True)
        return ops.PackedConv(weight.detach().float().transpose(0, 1).flip(2, 3).contiguous(), None, stride=1, padding=pad)
    pk = _cached(deps, ("pk", pad), make_pack)
    if stride != 1:
        Hd, Wd = H + 2 * pH - KH + 1, W + 2 * pW - KW + 1
        OH, OW = dy.shape[2:]
        g = torch.empty((B, Cout, Hd, Wd), dtype=torch.float32, device=dy.device)
        _check(lib.accflow_dilate_f32(_p(dy), _plane4(dy, "dy"), _p(g), B, Cout, OH, OW, Hd, Wd, int(stride), _stream()),
               "accflow_dilate_f32")
        dy = g
    with ops.conv_mode(GRAD_CONV_MODE):
        dx = ops.conv2d(pk, dy, out=out)
    if tuple(dx.shape[2:]) != (H, W):
        raise RuntimeError("conv_dgrad: got %s for an input of %s" % (tuple(dx.shape), (H, W)))
    return dx


def deform_conv_backward(x, offset, mask, weight, dy, need_dx=True, deps=None):
    """torchvision.ops.deform_conv2d (3x3, stride 1, pad 1, modulated; AccFlow_.py:104) backward ->
    (dx, doffset, dmask, dweight, dbias).  Columns cols[b][t*C + c] are re-formed (accflow_deform_columns_f32) for the
    weight gradient; dcols = W^T dy is a 1x1 convolution."""
    lib = _lib.load()
    B, C, H, W = x.shape
    Cout, _, KH, KW = weight.shape
    T = KH * KW
    pH, pW = (KH - 1) // 2, (KW - 1) // 2
    cols = torch.empty((B, T * C, H, W), dtype=torch.float32, device=x.device)
    _check(lib.accflow_deform_columns_f32(_p(x), _plane4(x, "x"), _p(offset), _plane4(offset, "offset"), _p(mask),
                                          _plane4(mask, "dmask"), _p(cols), B, C, H, W, KH, KW, pH, pW, _stream()),
           "accflow_deform_columns_f32")
    dwz, db = conv_wgrad(cols, dy, 1, 1)                                   # (Cout, T*C, 1, 1), column order t*C + c
    dweight = dwz.reshape(Cout, KH, KW, C).permute(0, 3, 1, 2).contiguous()
    dcols = conv_dgrad(dy, lambda: weight.detach().float().permute(0, 2, 3, 1).reshape(Cout, T * C, 1, 1), (H, W),
                       deps=deps)                                          # (B, T*C, H, W)
    dx = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    doff = torch.empty((B, 2 * T, H, W), dtype=torch.float32, device=x.device)
    dmsk = torch.empty((B, T, H, W), dtype=torch.float32, device=x.device)
    _check(lib.accflow_deform_conv_backward_f32(_p(x), _plane4(x, "x"), _p(offset), _plane4(offset, "offset"), _p(mask),
                                                _plane4(mask, "dmask"), _p(dcols), _p(dx), _plane4(dx, "dx"), _p(doff),
                                                _p(dmsk), B, C, H, W, KH, KW, pH, pW, _stream()),
           "accflow_deform_conv_backward_f32")
    return (dx if need_dx else None), doff, dmsk, dweight, db
