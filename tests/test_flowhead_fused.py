"""FlowHead as one convolution launch + the tap sum (ACCFLOW_EPI_TAPGEMM, ops.conv2d_tapgemm; update.py:12-13 of the
reference: conv2(relu(conv1(net)))): against float64 F.conv2d, against the three-launch form it replaces, the range guard,
the argument checks, and the estimator with the fusion on / off."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev(t):
    return t.cuda()


def maxerr(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max())


@pytest.fixture(scope="module")
def ops():
    from accflow_amd import ops as _ops
    assert torch.cuda.is_available()
    return _ops


def _head(seed, cin=128, mid=256, cout=2, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    w1 = torch.randn(mid, cin, 3, 3, generator=g) * (scale / (cin * 9) ** 0.5)
    b1 = torch.randn(mid, generator=g) * 0.1
    w2 = torch.randn(cout, mid, 3, 3, generator=g) / (mid * 9) ** 0.5
    b2 = torch.randn(cout, generator=g) * 0.1
    return w1, b1, w2, b2


def _ref64(x, w1, b1, w2, b2):
    y = F.relu(F.conv2d(x.double(), w1.double(), b1.double(), padding=1))
    return F.conv2d(y, w2.double(), b2.double(), padding=1)


@pytest.mark.parametrize("shape", [(5, 60, 128), (2, 124, 130), (17, 37, 53), (11, 60, 128)])
def test_tapgemm_vs_float64_and_three_launches(ops, shape):
    B, H, W = shape
    w1, b1, w2, b2 = _head(B)
    x = torch.randn(B, 128, H, W, generator=torch.Generator().manual_seed(100 + B))
    ref = _ref64(x, w1, b1, w2, b2)
    with ops.conv_mode("f16x3"):
        p1 = ops.PackedConv(dev(w1), dev(b1), padding=1)
        p2 = ops.PackedConv(dev(w2), dev(b2), padding=1)
        assert p2.ztaps_acc is not None
        x16 = ops.S16.from_float(dev(x))
        assert ops.tapgemm_eligible(p1, p2, x16)
        got = ops.conv2d_tapgemm(p1, p2, x16)
        # the form it replaces: conv1 -> S16 tensor -> 18-row 1x1 conv -> tap sum
        mid16 = ops.S16.empty(B, 256, H, W, dev(x).device)
        ops.conv2d(p1, x16, out16=mid16, act=ops.ACT_RELU, fp32_out=False)
        three = ops.conv2d(p2, mid16)
        # accumulate epilogue, and a replay through the descriptor cache
        base = dev(torch.randn(B, 2, H, W, generator=torch.Generator().manual_seed(7)))
        acc = base.clone()
        cache = {}
        ops.conv2d_tapgemm(p1, p2, x16, out=acc, epi=ops.EPI_ACCUM, e0=acc, cache=(cache, "k"))
        ops.conv2d_tapgemm(None, None, None, cache=(cache, "k"))
        assert not ops.guard_tripped()
    scale = float(ref.abs().max())
    e64 = maxerr(got, ref.float())
    e3 = maxerr(got, three)
    print("B%d %dx%d: vs float64 %.2e, vs three launches %.2e (max |ref| %.2f)" % (B, H, W, e64, e3, scale))
    assert e64 <= 2e-6 * scale + 1e-6, e64
    assert e3 <= 2e-6 * scale + 1e-6, e3                # same products; only the order of the fp32 sums over channels differs
    assert maxerr(acc, base + 2 * got) <= 4e-6 * scale + 2e-6
    again = None
    with ops.conv_mode("f16x3"):
        again = ops.conv2d_tapgemm(p1, p2, x16)
    assert torch.equal(again, got), "the result must be deterministic run to run"


def test_tapgemm_guard_and_argument_errors(ops):
    B, H, W = 5, 60, 128
    w1, b1, w2, b2 = _head(3, scale=400.0)           # conv1's outputs leave the fp16 split's range (|x| 2^4 >= 65520)
    x = torch.randn(B, 128, H, W, generator=torch.Generator().manual_seed(5)) * 30.0
    with ops.conv_mode("f16x3"):
        p1 = ops.PackedConv(dev(w1), dev(b1), padding=1)
        p2 = ops.PackedConv(dev(w2), dev(b2), padding=1)
        x16 = ops.S16.from_float(dev(x))
        ops.guard_tripped()
        ops.conv2d_tapgemm(p1, p2, x16)
        assert ops.guard_tripped(), "an activation beyond the scaled fp16 range must raise the flag"
        # shapes outside the fused form are refused loudly
        small = ops.S16.from_float(dev(x[:1]))
        assert not ops.tapgemm_eligible(p1, p2, small)            # one item of 60x128 = 120 workgroups: the split-K path's grid
        with pytest.raises(RuntimeError):
            ops.conv2d_tapgemm(p1, p2, small)
        w3 = torch.randn(4, 256, 3, 3)                             # 36 tap rows > 18
        p3 = ops.PackedConv(dev(w3), None, padding=1)
        assert p3.ztaps_acc is None and not ops.tapgemm_eligible(p1, p3, x16)
        with pytest.raises(RuntimeError):
            ops.conv2d_tapgemm(p1, p2, x16, epi=ops.EPI_ACCUM)     # no e0
    with ops.conv_mode("bf16x6"):
        assert not ops.tapgemm_eligible(p1, p2, x16)


def test_c_abi_rejects_bad_tapgemm_descriptors(ops):
    """accflow_conv2d_f32 with epi = ACCFLOW_EPI_TAPGEMM returns 1 (never launches) for descriptors outside the form."""
    import ctypes
    from accflow_amd import _lib
    lib = _lib.load()
    B, H, W = 5, 60, 128
    w1, b1, w2, b2 = _head(1)
    with ops.conv_mode("f16x3"):
        p1 = ops.PackedConv(dev(w1), dev(b1), padding=1)
        p2 = ops.PackedConv(dev(w2), dev(b2), padding=1)
        x16 = ops.S16.from_float(dev(torch.randn(B, 128, H, W)))
        cache = {}
        ops.conv2d_tapgemm(p1, p2, x16, cache=(cache, "k"))
        d = cache["k"][0]
        st = torch.cuda.current_stream().cuda_stream
        assert lib.accflow_conv2d_f32(ctypes.byref(d), st) == 0
        for field, bad in (("tg_rows", 19), ("tg_rows", 0), ("tg_coutpad", 16), ("act", ops.ACT_NONE), ("tg_out", None),
                           ("tg_w16", None), ("tg_out_bs", 17 * H * W), ("in_fmt", 0), ("mode", ops.CONV_BF16X6)):
            keep = getattr(d, field)
            setattr(d, field, bad)
            assert lib.accflow_conv2d_f32(ctypes.byref(d), st) == 1, field
            setattr(d, field, keep)
        assert lib.accflow_conv2d_f32(ctypes.byref(d), st) == 0
        z = torch.zeros(2, B, 18, H, W, device="cuda")
        o = torch.zeros(B, 2, H, W, device="cuda")
        f = lib.accflow_tap_sum_parts_f32
        assert f(z.data_ptr(), 0, z[0].numel(), 18 * H * W, None, None, 0, o.data_ptr(), 2 * H * W, B, 2, H, W, 3, 3, 1, 1, 0, 0, st) == 1
        assert f(z.data_ptr(), 2, 0, 18 * H * W, None, None, 0, o.data_ptr(), 2 * H * W, B, 2, H, W, 3, 3, 1, 1, 0, 0, st) == 1
        assert f(z.data_ptr(), 2, z[0].numel(), 18 * H * W, None, None, 0, o.data_ptr(), 2 * H * W, B, 2, H, W, 3, 3, 1, 1, 0, 4, st) == 1
        assert f(z.data_ptr(), 2, z[0].numel(), 18 * H * W, None, None, 0, o.data_ptr(), 2 * H * W, B, 2, H, W, 3, 3, 1, 1, 0, 0, st) == 0
    torch.cuda.synchronize()


def test_estimator_with_and_without_the_fused_flow_head(ops):
    """RAFT (batch 8 = two pair groups of 4 x 60x96 coarse pixels = 360 workgroups each, 6 iterations) with ACCFLOW_FUSE_FLOWHEAD on / off:
    the same flows up to fp32 rounding.  (The fused default is what the C2 / C3 / C5 reference-golden tests of
    test_hip_parity.py run.)"""
    from accflow_amd.data.synthetic import make_sequence, make_state_dict, normalize
    from accflow_amd.networks import build_flow_estimator
    from accflow_amd.networks.raft import update as U
    model = build_flow_estimator("raft")
    model.load_state_dict(make_state_dict(model), strict=True)
    model = model.cuda().eval()
    big = [dev(normalize(f)) for f in make_sequence(1005, 2, 480, 768, batch=8)]
    assert ops.FUSE_TAPGEMM
    flows, launches = {}, {}
    real = ops.conv2d_tapgemm
    try:
        for on in (True, False):
            ops.FUSE_TAPGEMM = on
            n = [0]

            def counted(*a, **k):
                n[0] += 1
                return real(*a, **k)
            ops.conv2d_tapgemm = counted
            with ops.conv_mode("f16x3"):
                flows[on] = model(big[1], big[0], iters=6).cpu()
            launches[on] = n[0]
    finally:
        ops.FUSE_TAPGEMM = True
        ops.conv2d_tapgemm = real
    assert launches == {True: 12, False: 0}, launches      # 2 pair groups x 6 iterations
    e = (flows[True] - flows[False]).pow(2).sum(1).sqrt()
    print("fused vs unfused flow head: EPE mean %.2e max %.2e (|flow| max %.1f)" % (float(e.mean()), float(e.max()), float(flows[True].abs().max())))
    assert 0.0 < float(e.mean()) <= 2e-5 and float(e.max()) <= 1e-3
