"""Known-answer vectors for torchvision.ops.deform_conv2d (modulated, v2) computed by an INDEPENDENT route.

The reference delegates AccPlus's deformable convolution to torchvision 0.16.1 (AccFlow_.py:4,83,104;
environment.yml:160), which is absent from this image, so the op cannot be executed here.  The oracle
(oracle/accflow_oracle.py::deform_conv2d) restates it with tensorised fp32 PyTorch code; this script is a
second, separately written restatement that shares no code with it: float64, one scalar loop nest per
(batch item, output pixel, tap) in the order of torchvision's CPU kernel
(torchvision/csrc/ops/cpu/deform_conv2d_kernel.cpp: `deformable_im2col_kernel` builds the column matrix
row (c_in*KH*KW + i*KW + j), `bilinear_interpolate` returns 0 for h <= -1 || h >= H || w <= -1 || w >= W and
otherwise sums the in-range corners; the output is weight.view(Cout, Cin*KH*KW) @ columns + bias).
It agrees by construction with the loop nest torchvision's own test-suite uses as its expected value
(test/test_ops.py::TestDeformConv.expected_fn: pi = stride*i - pad + dil*di + offset[b, 2*(di*KW+dj)],
pj = ... + offset[b, 2*(di*KW+dj)+1], out += mask * weight * bilinear(x, pi, pj)).

The offsets are built so that sample positions fall on every branch of the boundary rule: strictly inside,
exactly on integer pixels, inside the half-open bands (-1, 0) and (H-1, H) / (W-1, W) where only some corners
exist, exactly at -1 and at H / W (zero by the early-out), and far outside.

    python tests/golden/make_deform_golden.py      ->  tests/golden/deform_conv_kat.npz
"""
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def bilinear_f64(plane, H, W, h, w):
    """plane: (C, H, W) float64 - all channels of one batch item at once; (h, w) scalar sample position."""
    if h <= -1.0 or h >= H or w <= -1.0 or w >= W:
        return np.zeros(plane.shape[0], dtype=np.float64)
    h_low, w_low = int(math.floor(h)), int(math.floor(w))
    h_high, w_high = h_low + 1, w_low + 1
    lh, lw = h - h_low, w - w_low
    hh, hw = 1.0 - lh, 1.0 - lw
    acc = np.zeros(plane.shape[0], dtype=np.float64)
    if h_low >= 0 and w_low >= 0:
        acc += hh * hw * plane[:, h_low, w_low]
    if h_low >= 0 and w_high <= W - 1:
        acc += hh * lw * plane[:, h_low, w_high]
    if h_high <= H - 1 and w_low >= 0:
        acc += lh * hw * plane[:, h_high, w_low]
    if h_high <= H - 1 and w_high <= W - 1:
        acc += lh * lw * plane[:, h_high, w_high]
    return acc


def deform_conv2d_f64(x, offset, mask, weight, bias, pad=1):
    """3x3-style modulated deformable convolution, stride 1, dilation 1, one offset group, one weight group."""
    x, offset, mask = x.astype(np.float64), offset.astype(np.float64), mask.astype(np.float64)
    weight, bias = weight.astype(np.float64), bias.astype(np.float64)
    N, C, H, W = x.shape
    Cout, _, KH, KW = weight.shape
    OH, OW = H + 2 * pad - KH + 1, W + 2 * pad - KW + 1
    wmat = weight.reshape(Cout, C * KH * KW)
    out = np.zeros((N, Cout, OH, OW), dtype=np.float64)
    for b in range(N):
        for oy in range(OH):
            for ox in range(OW):
                col = np.zeros((C, KH * KW), dtype=np.float64)     # column (c, i*KW + j) of this output pixel
                for i in range(KH):
                    for j in range(KW):
                        t = i * KW + j
                        y = (oy - pad) + i + offset[b, 2 * t, oy, ox]
                        xx = (ox - pad) + j + offset[b, 2 * t + 1, oy, ox]
                        col[:, t] = mask[b, t, oy, ox] * bilinear_f64(x[b], H, W, y, xx)
                out[b, :, oy, ox] = wmat @ col.reshape(C * KH * KW) + bias
    return out


def make_case(seed, N, C, Cout, H, W):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    weight = (rng.standard_normal((Cout, C, 3, 3)) / math.sqrt(9 * C)).astype(np.float32)
    bias = (0.1 * rng.standard_normal(Cout)).astype(np.float32)
    mask = (1.0 / (1.0 + np.exp(-rng.standard_normal((N, 9, H, W))))).astype(np.float32)
    offset = (1.7 * rng.standard_normal((N, 18, H, W))).astype(np.float32)
    # steer selected (pixel, tap) samples onto the boundary branches: wanted absolute position -> offset
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    targets_h = [-1.0, -0.999, -0.5, -0.25, 0.0, H - 1.0, H - 0.75, H - 0.5, H - 0.001, float(H), H + 3.0, -4.0, 2.0, 1.5]
    targets_w = [-1.0, -0.999, -0.5, -0.25, 0.0, W - 1.0, W - 0.75, W - 0.5, W - 0.001, float(W), W + 3.0, -4.0, 3.0, 2.5]
    k = 0
    for b in range(N):
        for t in range(9):
            ky, kx = t // 3, t % 3
            sel = rng.random((H, W)) < 0.35
            th = np.array(targets_h)[(ys * 3 + xs + k) % len(targets_h)]
            tw = np.array(targets_w)[(ys + xs * 5 + 2 * k) % len(targets_w)]
            k += 1
            which = rng.integers(0, 3, size=(H, W))               # steer h only, w only, or both
            oh = (th - (ys - 1 + ky)).astype(np.float32)
            ow = (tw - (xs - 1 + kx)).astype(np.float32)
            offset[b, 2 * t][sel & (which != 1)] = oh[sel & (which != 1)]
            offset[b, 2 * t + 1][sel & (which != 0)] = ow[sel & (which != 0)]
    out = deform_conv2d_f64(x, offset, mask, weight, bias)
    return dict(x=x, offset=offset, mask=mask, weight=weight, bias=bias, out=out.astype(np.float32))


if __name__ == "__main__":
    g = {}
    # a: small and ragged; b: AccPlus's channel count (128 -> 128, AccFlow_.py:83) so that the matrix-core route runs
    for tag, args in (("a", (3, 2, 8, 6, 9, 11)), ("b", (5, 1, 128, 128, 12, 20))):
        for k, v in make_case(*args).items():
            g[tag + "_" + k] = v
    np.savez_compressed(os.path.join(HERE, "deform_conv_kat.npz"), **g)
    print({k: v.shape for k, v in g.items()})
