"""Known-answer vectors for the BACKWARD of torchvision.ops.deform_conv2d (modulated, v2), by an independent route.

torchvision 0.16.1 (AccFlow_.py:4,83,104; environment.yml:160) is absent from this image, so its binary cannot run here.
The training slice's deformable backward (csrc/backward.hip: deform_backward_kernel / deform_backward_lds_kernel) was so far
pinned only by autograd THROUGH the oracle's tensorised restatement of the forward (one restatement, differentiated).  This
script is a second, separately written statement of the gradients: float64, scalar loop nests in the structure of
torchvision's CPU backward (torchvision/csrc/ops/cpu/deform_conv2d_kernel.cpp), sharing no code with oracle/ or with
make_deform_golden.py's forward:

  * grad input  - `deformable_col2im_kernel`: the column gradient  W^T dY  of sample (c, tap, pixel) at position (y, x) is
    scattered to every INTEGER pixel (yp, xp) of the image with |y - yp| < 1 and |x - xp| < 1, weighted
    mask * (1 - |y - yp|) * (1 - |x - xp|)  (the kernel probes the 3 x 3 neighbourhood of the truncated position);
  * grad offset - `deformable_col2im_coord_kernel` with `get_coordinate_weight`: with y_l = floor(y), x_l = floor(x) and the
    four corner values v (zero for corners outside the image),  d/dy = dx (v_YX - v_yX) + (1 - dx)(v_Yx - v_yx),
    d/dx = dy (v_YX - v_Yx) + (1 - dy)(v_yX - v_yx),  each times mask * column gradient, summed over the channels;
  * grad mask   - the same kernel: column gradient times `bilinear_interpolate` (which returns 0 for y <= -1, y >= H,
    x <= -1, x >= W), summed over the channels;
  * grad weight = dY (Cout x pixels) @ columns^T with the modulated columns of the forward's `deformable_im2col_kernel`,
    grad bias = sum of dY.

Sample positions are steered into the boundary bands (-1, 0), (H - 1, H) / (W - 1, W), onto integer pixels and far outside;
the measure-zero positions exactly at -1 / H / W - where torchvision's two kernels disagree with each other by construction
(early-out in the value, one-sided difference in the coordinate weight) - are left out.

    python tests/golden/make_deform_backward_golden.py   ->  tests/golden/deform_conv_backward_kat.npz
"""
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def corner(img, H, W, yy, xx):
    """img: (H, W) float64; value of an integer corner, zero outside."""
    return img[yy, xx] if (0 <= yy < H and 0 <= xx < W) else 0.0


def sample_value(img, H, W, y, x):
    """bilinear_interpolate of one channel plane with torchvision's early-out."""
    if y <= -1.0 or y >= H or x <= -1.0 or x >= W:
        return 0.0
    y_l, x_l = int(math.floor(y)), int(math.floor(x))
    y_h, x_h = y_l + 1, x_l + 1
    ly, lx = y - y_l, x - x_l
    hy, hx = 1.0 - ly, 1.0 - lx
    v = 0.0
    if y_l >= 0 and x_l >= 0:
        v += hy * hx * img[y_l, x_l]
    if y_l >= 0 and x_h <= W - 1:
        v += hy * lx * img[y_l, x_h]
    if y_h <= H - 1 and x_l >= 0:
        v += ly * hx * img[y_h, x_l]
    if y_h <= H - 1 and x_h <= W - 1:
        v += ly * lx * img[y_h, x_h]
    return v


def coordinate_weight(img, H, W, y, x, along_y):
    y_l, x_l = int(math.floor(y)), int(math.floor(x))
    y_h, x_h = y_l + 1, x_l + 1
    v_yx, v_yX = corner(img, H, W, y_l, x_l), corner(img, H, W, y_l, x_h)
    v_Yx, v_YX = corner(img, H, W, y_h, x_l), corner(img, H, W, y_h, x_h)
    if along_y:
        dx = x - x_l
        return dx * (v_YX - v_yX) + (1.0 - dx) * (v_Yx - v_yx)
    dy = y - y_l
    return dy * (v_YX - v_Yx) + (1.0 - dy) * (v_yX - v_yx)


def deform_conv2d_backward_f64(x, offset, mask, weight, dy, pad=1):
    x, offset, mask, weight, dy = (a.astype(np.float64) for a in (x, offset, mask, weight, dy))
    N, C, H, W = x.shape
    Cout, _, KH, KW = weight.shape
    OH, OW = H + 2 * pad - KH + 1, W + 2 * pad - KW + 1
    T = KH * KW
    wmat = weight.reshape(Cout, C * T)                        # column row index = c * T + tap
    dx = np.zeros_like(x)
    doff = np.zeros_like(offset)
    dmask = np.zeros_like(mask)
    dw = np.zeros((Cout, C * T), dtype=np.float64)
    for b in range(N):
        for oy in range(OH):
            for ox in range(OW):
                g = dy[b, :, oy, ox]                          # (Cout)
                dcol = wmat.T @ g                             # (C * T): gradient of this pixel's column
                col = np.zeros(C * T, dtype=np.float64)       # the forward's modulated column (for grad weight)
                for i in range(KH):
                    for j in range(KW):
                        t = i * KW + j
                        y = (oy - pad) + i + offset[b, 2 * t, oy, ox]
                        xx = (ox - pad) + j + offset[b, 2 * t + 1, oy, ox]
                        m = mask[b, t, oy, ox]
                        g_y = g_x = g_m = 0.0
                        for c in range(C):
                            gc = dcol[c * T + t]
                            img = x[b, c]
                            val = sample_value(img, H, W, y, xx)
                            col[c * T + t] = m * val
                            g_m += gc * val
                            g_y += m * gc * coordinate_weight(img, H, W, y, xx, True)
                            g_x += m * gc * coordinate_weight(img, H, W, y, xx, False)
                            # col2im: the 3 x 3 integer neighbourhood of the truncated position
                            yt, xt = int(y), int(xx)          # C cast: truncation toward zero
                            for ddy in (-1, 0, 1):
                                for ddx in (-1, 0, 1):
                                    yp, xp = yt + ddy, xt + ddx
                                    if 0 <= yp < H and 0 <= xp < W and abs(y - yp) < 1.0 and abs(xx - xp) < 1.0:
                                        dx[b, c, yp, xp] += m * (1.0 - abs(y - yp)) * (1.0 - abs(xx - xp)) * gc
                        doff[b, 2 * t, oy, ox] = g_y
                        doff[b, 2 * t + 1, oy, ox] = g_x
                        dmask[b, t, oy, ox] = g_m
                dw += np.outer(g, col)
    return dx, doff, dmask, dw.reshape(weight.shape), dy.sum(axis=(0, 2, 3))


def make_case(seed, N, C, Cout, H, W):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    weight = (rng.standard_normal((Cout, C, 3, 3)) / math.sqrt(9 * C)).astype(np.float32)
    mask = (1.0 / (1.0 + np.exp(-rng.standard_normal((N, 9, H, W))))).astype(np.float32)
    offset = (1.7 * rng.standard_normal((N, 18, H, W))).astype(np.float32)
    dy = rng.standard_normal((N, Cout, H, W)).astype(np.float32)
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    th_all = np.array([-0.999, -0.5, -0.25, 0.0, 1.0, H - 1.0, H - 0.75, H - 0.5, H - 0.001, H + 3.0, -4.0, 2.0, 1.5, -1.5])
    tw_all = np.array([-0.999, -0.5, -0.25, 0.0, 2.0, W - 1.0, W - 0.75, W - 0.5, W - 0.001, W + 3.0, -4.0, 3.0, 2.5, W + 0.5])
    k = 0
    for b in range(N):
        for t in range(9):
            ky, kx = t // 3, t % 3
            sel = rng.random((H, W)) < 0.4
            th = th_all[(ys * 3 + xs + k) % len(th_all)]
            tw = tw_all[(ys + xs * 5 + 2 * k) % len(tw_all)]
            k += 1
            which = rng.integers(0, 3, size=(H, W))
            oh = (th - (ys - 1 + ky)).astype(np.float32)
            ow = (tw - (xs - 1 + kx)).astype(np.float32)
            offset[b, 2 * t][sel & (which != 1)] = oh[sel & (which != 1)]
            offset[b, 2 * t + 1][sel & (which != 0)] = ow[sel & (which != 0)]
    # keep every sample off the measure-zero positions exactly at -1 / H / W (see the module docstring)
    for t in range(9):
        ky, kx = t // 3, t % 3
        py = (ys - 1 + ky)[None] + offset[:, 2 * t].astype(np.float64)
        px = (xs - 1 + kx)[None] + offset[:, 2 * t + 1].astype(np.float64)
        offset[:, 2 * t][(py == -1.0) | (py == H)] += np.float32(0.37)
        offset[:, 2 * t + 1][(px == -1.0) | (px == W)] += np.float32(0.37)
    dx, doff, dmask, dw, db = deform_conv2d_backward_f64(x, offset, mask, weight, dy)
    return dict(x=x, offset=offset, mask=mask, weight=weight, dy=dy, dx=dx.astype(np.float32), doffset=doff.astype(np.float32),
                dmask=dmask.astype(np.float32), dweight=dw.astype(np.float32), dbias=db.astype(np.float32))


if __name__ == "__main__":
    g = {}
    # a: small and ragged; b: 16 channels at 12 x 20 (the LDS kernel's whole-plane form and the channel-group split);
    # c: a 72 x 64 plane (> 4096 pixels: the global-atomic fallback of deform_backward_kernel), 4 channels
    for tag, args in (("a", (3, 2, 6, 5, 7, 9)), ("b", (5, 1, 16, 8, 12, 20)), ("c", (7, 1, 4, 3, 72, 64))):
        for k, v in make_case(*args).items():
            g[tag + "_" + k] = v
    np.savez_compressed(os.path.join(HERE, "deform_conv_backward_kat.npz"), **g)
    print({k: v.shape for k, v in g.items()})
