"""GPU parity of the fused CorrBlock lookup -> convc1 kernel (csrc/corr_lookup_conv.hip, accflow_corr_lookup_convc1_s16):
relu(convc1(lookup)) as the oracle computes it - O.corr_lookup (raft/corr.py:24-45) followed by the motion encoder's first
convolution (raft/update.py:89-90) - on the same volume, coordinates and weights, incl. out-of-range / integer / non-finite
coordinates and a ragged last pixel tile; against the two-launch form of the same library; the range guard; and the whole
estimator with the fusion on and off."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import accflow_oracle as O

pytestmark = pytest.mark.gpu


def dev(t):
    return t.cuda()


def gen(seed):
    return torch.Generator().manual_seed(seed)


@pytest.fixture(scope="module")
def ops():
    from accflow_amd import ops as _ops
    assert torch.cuda.is_available()
    return _ops


def _case(shape, seed, noise=3.0):
    g = gen(seed)
    B, C, h, w = shape
    f1, f2 = torch.randn(*shape, generator=g), torch.randn(*shape, generator=g)
    pyr = O.corr_pyramid(f1, f2)
    coords = O.coords_grid(B, h, w) + noise * torch.randn(B, 2, h, w, generator=g)
    wgt = torch.randn(256, 324, 1, 1, generator=g) * 0.05
    wgt[7] *= 2.0 ** -9          # rows of very different magnitude: the pack's per-row power-of-two scales
    wgt[200] *= 2.0 ** 6
    bias = torch.randn(256, generator=g)
    return pyr, coords, wgt, bias


def _want(pyr, coords, wgt, bias, relu=True):
    c = torch.nan_to_num(coords, posinf=1e6, neginf=-1e6).clamp(-1e6, 1e6)
    y = F.conv2d(O.corr_lookup(pyr, c).double(), wgt.double(), bias.double())
    return (y.clamp(min=0) if relu else y).float()


def _fused(ops, pyr, coords, wgt, bias, shape, act=None, want16=True, want32=True):
    B, _, h, w = shape
    with ops.conv_mode("f16x3"):
        dp = ops.DispPyramid.from_rowmajor([dev(p) for p in pyr], B, h, w)
        pk = ops.PackedConv(ops.lookup_fused_weight(dev(wgt)), dev(bias))
        out16 = ops.S16.empty(B, 256, h, w, dp.levels[0].device, zero=True) if want16 else None
        out = torch.zeros((B, 256, h, w), dtype=torch.float32, device=dp.levels[0].device) if want32 else None
        ops.corr_lookup_convc1(dp, dev(coords), pk, out16=out16, out=out, act=ops.ACT_RELU if act is None else act)
    return out16, out


@pytest.mark.parametrize("shape", [(2, 32, 16, 24), (1, 32, 17, 22), (3, 32, 8, 8), (1, 16, 60, 128)])
def test_lookup_convc1_vs_oracle(ops, shape):
    """fp32 and pre-split outputs against float64 conv over the oracle's lookup; tolerance = the f16x3 convolution's
    (tests/test_s16m.py): max(5e-6, 2e-7 * sqrt(K)) of the output RMS, plus the lookup's own 2e-5-class blend rounding
    carried through 324 weights."""
    B, C, h, w = shape
    pyr, coords, wgt, bias = _case(shape, 31)
    coords[0, :, 0, 0] = torch.tensor([-7.5, 3.25])                    # a window partly outside
    coords[0, :, 1, 1] = torch.tensor([4.0, 5.0])                      # integer coordinates
    coords[0, :, 2, :4] = torch.tensor([[-30.0, 1e5, float(w) + 2.25, -1e9], [2.0, 3.0, float(h) - 0.5, 1e9]])
    coords[0, :, 3, :2] = torch.tensor([[float("inf"), float(w - 1)], [0.0, float(h - 1)]])
    want = _want(pyr, coords, wgt, bias)
    ops.guard_tripped()
    out16, out = _fused(ops, pyr, coords, wgt, bias, shape)
    assert not ops.guard_tripped()
    # per output channel: the convolution's own bound on the channel's RMS + the lookup's blend rounding (<= 2e-5 per tap
    # against the oracle, independent from tap to tap) carried through that channel's weights
    rms = want.pow(2).mean(dim=(0, 2, 3)).sqrt()
    tol = (max(5e-6, 2e-7 * math.sqrt(324)) * rms + 3 * 2e-5 * wgt.reshape(256, 324).norm(dim=1)).view(1, 256, 1, 1)
    for name, got in (("fp32", out.cpu()), ("S16", out16.to_float().cpu())):
        err = (got - want).abs()
        assert bool((err <= tol).all()), "fused lookup+convc1 %s %s: max err %.3e (worst ratio to tol %.2f)" % (
            name, shape, float(err.max()), float((err / tol).max()))
    # without the activation, and each destination alone
    want_lin = _want(pyr, coords, wgt, bias, relu=False)
    _, lin = _fused(ops, pyr, coords, wgt, bias, shape, act=ops.ACT_NONE, want16=False)
    assert bool(((lin.cpu() - want_lin).abs() <= tol).all())
    only16, _ = _fused(ops, pyr, coords, wgt, bias, shape, want32=False)
    assert torch.equal(only16.data, out16.data)


def test_lookup_convc1_equals_two_launches(ops):
    """The fused kernel blends and splits the taps exactly as accflow_corr_lookup_disp_s16 and multiplies them by the same
    fp16 weight terms as convc1's S16 pack; only the order of the fp32 accumulation differs (k = (super-step, level, tap)
    instead of (level, tap)): equal up to fp32 association of 324 products."""
    shape = (2, 32, 16, 24)
    B, C, h, w = shape
    pyr, coords, wgt, bias = _case(shape, 33)
    out16, _ = _fused(ops, pyr, coords, wgt, bias, shape, want32=False)
    with ops.conv_mode("f16x3"):
        dp = ops.DispPyramid.from_rowmajor([dev(p) for p in pyr], B, h, w)
        l16 = ops.S16.empty(B, ops.LOOKUP_S16_CHANNELS, h, w, dp.levels[0].device, zero=True)
        ops.corr_lookup_s16(dp, dev(coords), l16)
        w88 = torch.zeros(256, 4, 88)
        w88[:, :, :81] = wgt.reshape(256, 4, 9, 9).transpose(2, 3).reshape(256, 4, 81)
        pk = ops.PackedConv(dev(w88.reshape(256, 352, 1, 1)), dev(bias))
        two16 = ops.S16.empty(B, 256, h, w, dp.levels[0].device, zero=True)
        ops.conv2d(pk, l16, out16=two16, act=ops.ACT_RELU, fp32_out=False)
    a, b = out16.to_float().cpu(), two16.to_float().cpu()
    # per output channel (rows differ by 2^15 in magnitude): a few fp32 roundings of the channel's typical partial sum
    rms = b.pow(2).mean(dim=(0, 2, 3)).sqrt().view(1, 256, 1, 1)
    err = (a - b).abs()
    pre = b.abs().amax(dim=(0, 2, 3)).view(1, 256, 1, 1) + float(bias.abs().max())   # (scale of the pre-activation sums)
    assert bool((err <= 2e-6 * rms + 1e-6 * pre).all()), float((err / (2e-6 * rms + 1e-6 * pre)).max())


def test_lookup_convc1_guard_and_errors(ops):
    """A tap beyond the scaled fp16 range (|x| * 2^4 >= 65520) trips the guard; unsupported arguments are rejected."""
    shape = (1, 32, 16, 24)
    pyr, coords, wgt, bias = _case(shape, 35, noise=0.5)
    pyr = [p.clone() for p in pyr]
    pyr[2][100, 0, 1, 2] = 5000.0
    coords[0, :, 100 // 24, 100 % 24] = torch.tensor([2.0 * 4, 1.0 * 4])     # integer position (2, 1) of level 2
    ops.guard_tripped()
    _fused(ops, pyr, coords, wgt, bias, shape)
    assert ops.guard_tripped()
    pyr[2][100, 0, 1, 2] = float("nan")
    _fused(ops, pyr, coords, wgt, bias, shape)
    assert ops.guard_tripped()
    with ops.conv_mode("f16x3"):
        dp = ops.DispPyramid.from_rowmajor([dev(p) for p in pyr], 1, 16, 24)
        with pytest.raises(RuntimeError):        # not the fused pack (324 channels)
            ops.corr_lookup_convc1(dp, dev(coords), ops.PackedConv(dev(wgt), dev(bias)),
                                   out=torch.zeros(1, 256, 16, 24, device="cuda"))
        with pytest.raises(RuntimeError):        # no destination
            ops.corr_lookup_convc1(dp, dev(coords), ops.PackedConv(ops.lookup_fused_weight(dev(wgt)), dev(bias)))


def test_estimator_fused_vs_two_launches(ops):
    """RAFT forward with the fusion on / off (ACCFLOW_FUSE_LOOKUP): the same flows up to fp32 rounding, both within the
    north-star gate of the oracle."""
    from accflow_amd.data.synthetic import make_sequence, make_state_dict, normalize
    from accflow_amd.networks import build_flow_estimator
    from accflow_amd.networks.raft import update as U
    model = build_flow_estimator("raft")
    sd = make_state_dict(model)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    frames = [normalize(f) for f in make_sequence(1003, 2, 128, 256)]
    saved = U.FUSE_LOOKUP
    try:
        flows = {}
        for on in (True, False):
            U.FUSE_LOOKUP = on
            model.update_block._packs.clear()
            flows[on] = model(dev(frames[1]), dev(frames[0]), iters=4).cpu()
    finally:
        U.FUSE_LOOKUP = saved
    ref = O.raft_forward(sd, frames[1], frames[0], iters=4)
    for on in (True, False):
        m, mx = O.epe(flows[on], ref)
        assert m <= 1e-3, (on, m, mx)
    m, mx = O.epe(flows[True], flows[False])
    assert m <= 2e-5, (m, mx)
