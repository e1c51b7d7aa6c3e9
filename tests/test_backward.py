"""Backward operators of the fusion heads (accflow_amd/backward.py, csrc/backward.hip) against torch autograd
of the SAME operator evaluated in float64 on the CPU - the arithmetic the reference's training step differentiates
(train_acc.py:222-229).  Tolerance: 1e-4 of the gradient's RMS (fp32 sums of up to ~10^5 products in an unspecified order)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-4


def gen(seed):
    return torch.Generator().manual_seed(seed)


def rel(got, want):
    want = want.float()
    rms = float(want.pow(2).mean().sqrt().clamp_min(1e-30))
    return float((got.cpu() - want).abs().max()) / rms


@pytest.fixture(scope="module")
def bw():
    from accflow_amd import backward
    return backward


CONV_CASES = [
    # B, Cin, Cout, H, W, K, stride, pad
    (2, 128, 256, 12, 20, 3, 1, 1),     # AccPlus / FlowDecoder 3x3
    (1, 257, 256, 9, 13, 3, 1, 1),      # conv1[0]: 2c + 1 inputs
    (3, 2, 128, 10, 14, 7, 1, 3),       # FlowEncoder.conv1
    (2, 256, 128, 8, 8, 1, 1, 0),       # 1x1
    (2, 256, 2, 11, 15, 3, 1, 1),       # flow head: 2 outputs
    (2, 256, 1, 11, 15, 3, 1, 1),       # Blending mask: 1 output
    (2, 3, 64, 20, 28, 7, 2, 3),        # context stem, stride 2
    (2, 64, 96, 10, 14, 3, 2, 1),       # layer2 conv1, stride 2
    (2, 64, 96, 11, 15, 1, 2, 0),       # downsample 1x1 stride 2, odd size
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_backward_vs_autograd(bw, case):
    B, Cin, Cout, H, W, K, s, p = case
    g = gen(hash(case) % 1000)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    x64, w64, b64 = (t.double().requires_grad_() for t in (x, w, b))
    y = F.conv2d(x64, w64, b64, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    dw, db = bw.conv_wgrad(x.cuda(), dy.cuda(), K, K, stride=s, padding=(p, p))
    dx = bw.conv_dgrad(dy.cuda(), w.cuda(), (H, W), stride=s, padding=(p, p))
    assert rel(dw, w64.grad) < TOL
    assert rel(db, b64.grad) < TOL
    assert rel(dx, x64.grad) < TOL


# Shapes of the branch-free weight-gradient kernel (conv_wgrad_mfma_fast_kernel: stride 1, OW == W, W % 8 == 0, |kx - padW| <= 1),
# chosen to hit its edges: tile rows / columns past Cout / J, a pixel count that is no multiple of the 16-pixel step, items of a
# single 8-pixel chunk per row, one and two output channels, one-sided kernels, tall kernels with two padding rows.
FAST_WGRAD_CASES = [
    # B, Cin, Cout, H, W, KH, KW, padH, padW
    (2, 128, 256, 12, 16, 3, 3, 1, 1),
    (3, 257, 200, 9, 24, 3, 3, 1, 1),      # J = 2313, Cout = 200: ragged tiles both ways
    (3, 64, 64, 3, 8, 3, 3, 1, 1),         # 24 pixels per item, 72 in all: 4.5 steps; every chunk touches both row ends
    (2, 96, 96, 16, 32, 3, 3, 1, 1),
    (2, 256, 2, 10, 16, 3, 3, 1, 1),
    (2, 256, 1, 10, 16, 3, 3, 1, 1),
    (5, 40, 130, 7, 40, 1, 3, 0, 1),       # 1 x 3
    (2, 40, 72, 9, 16, 3, 1, 1, 0),        # 3 x 1
    (2, 24, 48, 11, 8, 5, 1, 2, 0),        # 5 x 1: two padding rows above and below
    (7, 1152, 128, 4, 8, 1, 1, 0, 0),      # the deformable convolution's 1x1 over its columns
    (2, 16, 32, 6, 16, 2, 3, 0, 1),        # even kernel height: OH = H - 1 (only the width must be "same")
    # (Cout <= 64 runs the 64 x 256 tile, two im2col rows per thread - also the 64 / 48 / 2 / 1-channel cases above)
    (2, 64, 64, 16, 32, 3, 3, 1, 1),       # J = 576: two full column tiles of 256 and a ragged third
    (2, 130, 27, 9, 16, 3, 3, 1, 1),       # J = 1170, 27 rows: ragged both ways; rows 128.. of a column tile hold other channels
    (3, 20, 64, 5, 24, 1, 3, 0, 1),        # J = 60: a single, mostly empty column tile
]


@pytest.mark.parametrize("case", FAST_WGRAD_CASES)
def test_conv_wgrad_fast_kernel_shapes(bw, case):
    B, Cin, Cout, H, W, KH, KW, pH, pW = case
    g = gen(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g)
    w64 = torch.zeros(Cout, Cin, KH, KW, dtype=torch.double, requires_grad=True)
    b64 = torch.zeros(Cout, dtype=torch.double, requires_grad=True)
    y = F.conv2d(x.double(), w64, b64, padding=(pH, pW))
    assert y.shape[3] == W
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    dw, db = bw.conv_wgrad(x.cuda(), dy.cuda(), KH, KW, padding=(pH, pW))
    assert rel(dw, w64.grad) < TOL, rel(dw, w64.grad)
    assert rel(db, b64.grad) < TOL
    again, _ = bw.conv_wgrad(x.cuda(), dy.cuda(), KH, KW, padding=(pH, pW))
    assert rel(again, dw.cpu()) < 1e-5        # (float atomics over the pixel parts: the order of the sums may differ)


@pytest.mark.parametrize("Cout", [40, 72])
@pytest.mark.parametrize("W", [8, 16, 40])
def test_conv_wgrad_fast_kernel_border_columns(bw, W, Cout):
    """the end-to-end gradient fixtures gate single elements at 2 % of the RMS, which a localized border bug of
    conv_wgrad_mfma_fast_kernel could pass.  Here dY lives ONLY in the first and the last 8-pixel chunk's border columns
    (ox = 0 and ox = OW - 1), so every product of the gradient involves a chunk whose kx = 0 / kx = 2 rows overhang the
    image row (zero padding on one side, the NEIGHBOURING row's pixel in memory on the other): per element against float64
    autograd at 1e-5 of the RMS, and the kx = 0 / 2 taps separately (a wrong neighbour pixel lands in exactly those)."""
    g = gen(100 + W)
    B, Cin, H = 2, 48, 6          # (Cout = 40: the 64 x 256 tile, 72: the 128 x 128 tile)
    x = torch.randn(B, Cin, H, W, generator=g)
    dy = torch.zeros(B, Cout, H, W)
    dy[..., 0] = torch.randn(B, Cout, H, generator=g)
    dy[..., W - 1] = torch.randn(B, Cout, H, generator=g)
    w64 = torch.zeros(Cout, Cin, 3, 3, dtype=torch.double, requires_grad=True)
    F.conv2d(x.double(), w64, None, padding=1).backward(dy.double())
    dw, db = bw.conv_wgrad(x.cuda(), dy.cuda(), 3, 3, padding=(1, 1))
    assert rel(dw, w64.grad) < 1e-5, rel(dw, w64.grad)
    for kx in (0, 2):
        assert rel(dw[..., kx], w64.grad[..., kx]) < 1e-5, kx
    assert rel(db, dy.double().sum((0, 2, 3))) < 1e-5


def test_conv_wgrad_on_channel_slices(bw):
    """Operands that are channel slices of wider buffers (the concatenation layouts of AccPlus, AccFlow_.py:98-107)."""
    g = gen(5)
    X, DY = torch.randn(2, 40, 9, 11, generator=g), torch.randn(2, 50, 9, 11, generator=g)
    x, dy = X[:, 8:32], DY[:, 10:42]
    w = torch.zeros(32, 24, 3, 3, dtype=torch.double, requires_grad=True)
    F.conv2d(x.double(), w, None, padding=1).backward(dy.double())
    dw, db = bw.conv_wgrad(X.cuda()[:, 8:32], DY.cuda()[:, 10:42], 3, 3, padding=(1, 1))
    assert rel(dw, w.grad) < TOL and rel(db, dy.double().sum((0, 2, 3))) < TOL
    # the same through the branch-free kernel (W % 8 == 0): batch strides that differ from the channel count
    X, DY = torch.randn(3, 40, 6, 16, generator=g), torch.randn(3, 50, 6, 16, generator=g)
    x, dy = X[:, 8:32], DY[:, 10:42]
    w = torch.zeros(32, 24, 3, 3, dtype=torch.double, requires_grad=True)
    F.conv2d(x.double(), w, None, padding=1).backward(dy.double())
    dw, db = bw.conv_wgrad(X.cuda()[:, 8:32], DY.cuda()[:, 10:42], 3, 3, padding=(1, 1))
    assert rel(dw, w.grad) < TOL and rel(db, dy.double().sum((0, 2, 3))) < TOL


@pytest.mark.parametrize("act", ["relu", "sigmoid", "tanh"])
def test_act_backward(bw, act):
    from accflow_amd import ops
    g = gen(7)
    pre = torch.randn(2, 27, 6, 9, generator=g, dtype=torch.double, requires_grad=True)
    y = getattr(torch, act)(pre)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    code = {"relu": ops.ACT_RELU, "sigmoid": ops.ACT_SIGMOID, "tanh": ops.ACT_TANH}[act]
    # slices: the sigmoid of AccPlus applies to channels 18..26 of a 27-channel tensor (AccFlow_.py:102-103)
    got = bw.act_backward(dy.cuda()[:, 18:], y.detach().float().cuda()[:, 18:], code)
    assert rel(got, pre.grad[:, 18:]) < 1e-5


def test_blend_backward(bw):
    g = gen(9)
    f1, f2 = (torch.randn(2, 16, 5, 7, generator=g, dtype=torch.double, requires_grad=True) for _ in range(2))
    m = torch.rand(2, 1, 5, 7, generator=g, dtype=torch.double, requires_grad=True)
    dy = torch.randn(2, 16, 5, 7, generator=g)
    (f1 * m + (1 - m) * f2).backward(dy.double())
    d1, d2, dm = bw.blend_backward(dy.cuda(), f1.detach().float().cuda(), f2.detach().float().cuda(), m.detach().float().cuda())
    assert rel(d1, f1.grad) < 1e-5 and rel(d2, f2.grad) < 1e-5 and rel(dm, m.grad) < 1e-5


def test_convex_upsample_backward(bw):
    """raft.py:81-92 written with torch ops in float64 (the oracle's convex_upsample is the same formula)."""
    g = gen(11)
    N, H, W = 2, 6, 9
    flow = torch.randn(N, 2, H, W, generator=g, dtype=torch.double, requires_grad=True)
    mask = torch.randn(N, 576, H, W, generator=g, dtype=torch.double, requires_grad=True)
    m = torch.softmax(mask.view(N, 1, 9, 8, 8, H, W), dim=2)
    up = F.unfold(8 * flow, [3, 3], padding=1).view(N, 2, 9, 1, 1, H, W)
    up = torch.sum(m * up, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(N, 2, 8 * H, 8 * W)
    dup = torch.randn(up.shape, generator=g)
    up.backward(dup.double())
    dflow, dmask = bw.convex_upsample_backward(dup.cuda(), flow.detach().float().cuda(), mask.detach().float().cuda())
    assert rel(dflow, flow.grad) < 1e-5 and rel(dmask, mask.grad) < 1e-5


def test_l1_grad_and_add(bw):
    g = gen(13)
    p, t = torch.randn(2, 2, 8, 8, generator=g), torch.randn(2, 2, 8, 8, generator=g)
    p[0, 0, 0, 0] = t[0, 0, 0, 0]
    got = bw.l1_grad(p.cuda(), t.cuda(), 1.0 / p.numel()).cpu()
    assert torch.equal(got, torch.sign(p - t) / p.numel())
    a, b = torch.randn(2, 6, 4, 4, generator=g), torch.randn(2, 3, 4, 4, generator=g)
    A = a.cuda()
    bw.add_(A[:, 2:5], b.cuda())
    a[:, 2:5] += b
    assert torch.equal(A.cpu(), a)


@pytest.mark.parametrize("big_offsets", [False, True])
def test_deform_conv_backward_vs_oracle_autograd(bw, big_offsets):
    """All five gradients of the modulated deformable convolution against autograd through the oracle's restatement
    (oracle.deform_conv2d, pinned by tests/golden/deform_conv_kat.npz) in float64; with offsets that leave the image."""
    from oracle import accflow_oracle as O
    g = gen(17 + big_offsets)
    B, C, Cout, H, W = 2, 24, 20, 7, 10
    x = torch.randn(B, C, H, W, generator=g)
    off = torch.randn(B, 18, H, W, generator=g) * (4.0 if big_offsets else 0.7)
    m = torch.rand(B, 9, H, W, generator=g)
    w = torch.randn(Cout, C, 3, 3, generator=g) / (9 * C) ** 0.5
    b = torch.randn(Cout, generator=g)
    t64 = [t.double().requires_grad_() for t in (x, off, m, w, b)]
    y = O.deform_conv2d(*t64)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    dx, doff, dm, dw, db = bw.deform_conv_backward(x.cuda(), off.cuda(), m.cuda(), w.cuda(), dy.cuda())
    for got, want in zip((dx, doff, dm, dw, db), t64):
        assert rel(got, want.grad) < TOL


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_deform_conv_backward_vs_independent_known_answers(bw, tag):
    """The five gradients against tests/golden/deform_conv_backward_kat.npz - float64 scalar loops after torchvision's CPU
    backward kernels, written independently of the oracle (tests/golden/make_deform_backward_golden.py).  a: ragged 7 x 9;
    b: 16 channels = one channel group of the LDS kernel; c: a 72 x 64 plane (> 4096 pixels: the global-atomic fallback)."""
    import os
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "deform_conv_backward_kat.npz"))
    x, off, m, w, dy = (torch.from_numpy(g[tag + "_" + k]).cuda() for k in ("x", "offset", "mask", "weight", "dy"))
    got = bw.deform_conv_backward(x, off, m, w, dy)
    for t, name in zip(got, ("dx", "doffset", "dmask", "dweight", "dbias")):
        e = rel(t, torch.from_numpy(g[tag + "_" + name]))
        assert e < TOL, (tag, name, e)


@pytest.mark.parametrize("case", [(64, 40, 3, 3, None, False), (130, 24, 1, 1, None, False), (27, 128, 3, 3, None, False),
                                  (2, 256, 3, 3, None, False), (96, 3, 7, 7, None, False), (128, 384, 1, 5, 256, False),
                                  (128, 128, 3, 3, None, True), (40, 64, 3, 3, None, "scale")])
def test_fused_pack_equals_the_single_purpose_packs(case):
    """accflow_conv_pack_all_f32 (one launch per convolution; the training step rebuilds ~80 packs per step) writes every pack
    bit-identically to the single-purpose entry points; its transpose_flip form equals the pack of the transposed, flipped copy."""
    from accflow_amd import ops
    Cout, Cin, KH, KW, C0, variant = case
    g = gen(Cout + Cin)
    w = (torch.randn(Cout, Cin, KH, KW, generator=g) * torch.logspace(-3, 1, Cout).view(-1, 1, 1, 1)).cuda()
    w[min(3, Cout - 1)] = 0.0                                    # an all-zero row: scale exponent 0
    sc = (torch.rand(Cout, generator=g) + 0.5).cuda() if variant == "scale" else None
    tap_major = variant is True

    def packs(fused, **kw):
        saved = ops.PACK_FUSED
        ops.PACK_FUSED = fused
        try:
            return ops.PackedConv(kw.pop("weight", w), None, padding=(KH // 2, KW // 2), scale=sc, C0=C0, tap_major=tap_major, **kw)
        finally:
            ops.PACK_FUSED = saved

    a, b = packs(True), packs(False)
    for name in ("wpack", "ktab", "wsplit", "wpatch", "wpatch16", "wscale16", "wsplit16"):
        x, y = getattr(a, name), getattr(b, name)
        assert (x is None) == (y is None), name
        if x is None:
            continue
        if name in ("wpatch16", "wsplit16"):    # the third term's slot is never read (and not written by every packer)
            n3 = x.numel() // 3
            x, y = x[:2 * n3], y[:2 * n3]
        assert torch.equal(x, y), (case, name)
    if not tap_major and sc is None and Cin > 4:
        t = packs(True, weight=w, transpose_flip=True)           # logical weight: (Cin, Cout, KH, KW)
        u = ops.PackedConv(w.transpose(0, 1).flip(2, 3).contiguous(), None, padding=(KH // 2, KW // 2))
        assert (t.Cout, t.Cin) == (Cin, Cout) == (u.Cout, u.Cin)
        for name in ("wpack", "ktab", "wsplit", "wpatch", "wpatch16", "wscale16", "wsplit16"):
            x, y = getattr(t, name), getattr(u, name)
            assert (x is None) == (y is None), name
            if x is not None:
                if name in ("wpatch16", "wsplit16"):
                    n3 = x.numel() // 3
                    x, y = x[:2 * n3], y[:2 * n3]
                assert torch.equal(x, y), (case, "transpose_flip", name)
