"""The multi-source S16 convolution (accflow_conv_desc.nsrc, csrc/conv_s16m_kernel.h): channel concatenations of up to four
pre-split tensors without materialising the cat (AccPlus, AccFlow_.py:98-107), stride-2 convolutions as stride-1 work over
the input's pixel-parity classes (extractor.py:9,52), and the kernel's wave layouts.  Checked against the CPU fp32
convolution (the oracle's arithmetic: torch.nn.functional.conv2d on the CPU) on seeded inputs incl. ragged sizes, and -
where the summation order is the same - for bit-identity against the library's fp32-activation path."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev(t):
    return t.cuda()


@pytest.fixture(scope="module")
def ops():
    from accflow_amd import ops as o
    if o.conv_mode_name() != "f16x3":
        pytest.skip("S16 tensors belong to the f16x3 mode")
    return o


def gen(seed):
    return torch.Generator().manual_seed(seed)


@pytest.fixture
def no_ksplit(ops, monkeypatch):
    """The wave layouts differ in how many workgroups a launch has, hence in whether a small launch is split along K (another
    summation order): layout-vs-layout bit comparisons run without the split-K workspace."""
    monkeypatch.setattr(ops, "USE_KSPLIT", False)


def tol(K):
    """fp32 accumulation of K products: the rounding error grows like sqrt(K) * 2^-24 of the output RMS (5e-6 up to K ~ 600)"""
    return max(5e-6, 2e-7 * K ** 0.5)


def rel_err(got, want):
    """max error relative to the per-channel output RMS (tests/test_hip_parity.py's conv metric)"""
    rms = want.pow(2).mean(dim=(0, 2, 3), keepdim=True).sqrt().clamp_min(1e-30)
    return float(((got - want).abs() / rms).max())


CAT_CASES = [
    # member channels, Cout, KH, KW, B, H, W
    ((128, 128, 1), 256, 3, 3, 1, 60, 128),      # AccPlus conv1[0] / conv3[0]: cat[df, f, o]
    ((128, 128, 128, 128), 256, 3, 3, 1, 60, 128),   # AccPlus conv4[0]: cat[x, c, f_, df]
    ((128, 128), 256, 3, 3, 2, 13, 37),          # ragged tiles
    ((40, 24, 9), 64, 3, 3, 2, 17, 45),          # members that are no multiple of 16: each is padded on its own
    ((48, 80), 96, 1, 1, 2, 16, 64),             # 1x1 with a 16-channel tail group in the first member
    ((16, 3, 128), 27, 1, 5, 1, 9, 33),
]


@pytest.mark.parametrize("case", CAT_CASES)
def test_multi_source_cat_vs_cpu_conv(ops, case, no_ksplit):
    cs, Cout, KH, KW, B, H, W = case
    g = gen(11)
    xs = [torch.randn(B, c, H, W, generator=g) for c in cs]
    Cin = sum(cs)
    w = torch.randn(Cout, Cin, KH, KW, generator=g) * (1.0 / (Cin * KH * KW)) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    want = F.conv2d(torch.cat(xs, 1).double(), w.double(), b.double(), padding=(KH // 2, KW // 2)).float()
    pk = ops.PackedMulti.from_cat(dev(w), dev(b), list(cs), (KH // 2, KW // 2))
    x16 = [ops.to_s16(dev(x)) for x in xs]
    got = ops.conv2d_multi(pk, x16)
    assert rel_err(got.cpu(), want) <= tol(Cin * KH * KW)
    # every wave layout computes the same sums in the same order
    for lay in (0, 1, 2, 3):
        assert torch.equal(ops.conv2d_multi(pk, x16, lay=lay), got), lay
    # S16 output = the split of the fp32 output; relu epilogue
    o16 = ops.S16.empty(B, Cout, H, W, got.device, zero=True)
    r = ops.conv2d_multi(pk, x16, act=ops.ACT_RELU, out16=o16, fp32_out=False)
    assert r is o16
    ref16 = ops.to_s16(torch.relu(got)).to_float()
    assert torch.equal(o16.to_float(), ref16)
    assert not ops.guard_tripped()


def test_multi_source_equals_two_source_path_bitwise(ops):
    """Members whose channel counts are multiples of 16 (all but the last): the reduction order is the one of the conv over
    the materialised cat, so the multi-source result equals the fp32-activation path of the library bit for bit."""
    g = gen(12)
    B, H, W = 1, 60, 128
    cs = (128, 128, 1)
    xs = [dev(torch.randn(B, c, H, W, generator=g)) for c in cs]
    w = dev(torch.randn(256, 257, 3, 3, generator=g) * 0.02)
    b = dev(torch.randn(256, generator=g) * 0.1)
    cat = torch.cat(xs, 1)
    want = ops.conv2d(ops.PackedConv(w, b, padding=1), cat, act=ops.ACT_RELU)
    pk = ops.PackedMulti.from_cat(w, b, list(cs), 1)
    got = ops.conv2d_multi(pk, [ops.to_s16(x) for x in xs], act=ops.ACT_RELU)
    assert torch.equal(got, want)


STRIDED = [
    # Cin, Cout, K, pad, B, H, W
    (64, 96, 3, 1, 2, 48, 96),        # layer2 conv1 (extractor.py:9)
    (64, 96, 1, 0, 2, 48, 96),        # layer2 downsample (extractor.py:52)
    (96, 128, 3, 1, 1, 30, 64),
    (96, 128, 1, 0, 1, 30, 64),
    (32, 64, 3, 1, 2, 17, 41),        # odd sizes: the last row / column has no right / lower neighbour
    (16, 64, 7, 3, 1, 20, 36),        # 7x7: classes of 3 and 4 taps per axis
    (24, 40, 5, 2, 1, 14, 70),
]


@pytest.mark.parametrize("case", STRIDED)
def test_stride2_as_parity_sources_vs_cpu_conv(ops, case, no_ksplit):
    Cin, Cout, K, p, B, H, W = case
    g = gen(13)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) * (1.0 / (Cin * K * K)) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    want = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=p).float()
    pk = ops.PackedMulti.from_strided(dev(w), dev(b), p)
    x16 = ops.to_s16(dev(x))
    got = ops.conv2d_multi(pk, [x16] * len(pk.C), out_hw=want.shape[2:])
    assert tuple(got.shape) == tuple(want.shape)
    assert rel_err(got.cpu(), want) <= 5e-6
    for lay in (0, 1, 2, 3):
        assert torch.equal(ops.conv2d_multi(pk, [x16] * len(pk.C), out_hw=want.shape[2:], lay=lay), got), lay
    assert not ops.guard_tripped()


@pytest.mark.parametrize("lay", [0, 1, 2, 3])
def test_statistics_of_an_s16_convolution(ops, lay):
    """InstanceNorm statistics gathered by the multi-source kernel's epilogue (S16 input, raw fp32 output): the norm
    applied from them equals the oracle's instance norm of the convolution."""
    g = gen(14)
    B, Cin, Cout, H, W = 2, 64, 96, 24, 64
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g) * 0.1
    y = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    want = torch.relu(F.instance_norm(y, eps=1e-5)).float()
    pk = ops.PackedMulti.from_cat(dev(w), dev(b), [Cin], 1)
    got, st = ops.conv2d_multi(pk, [ops.to_s16(dev(x))], want_stats=True, lay=lay)
    assert st is not None
    ops.instance_norm(got, 1, stats=st)
    assert float((got.cpu() - want).abs().max()) <= 2e-5


def test_split_k_and_residual_epilogues(ops):
    """Batch-1 launches (the fusion chain) split the reduction over the sources' chunks; residual / accumulate epilogues."""
    g = gen(15)
    B, H, W = 1, 60, 128
    cs = (128, 128, 128, 128)
    xs = [torch.randn(B, c, H, W, generator=g) for c in cs]
    w = torch.randn(128, 512, 3, 3, generator=g) * 0.01
    b = torch.randn(128, generator=g) * 0.1
    res = torch.randn(B, 128, H, W, generator=g)
    conv = F.conv2d(torch.cat(xs, 1).double(), w.double(), b.double(), padding=1).float()
    pk = ops.PackedMulti.from_cat(dev(w), dev(b), list(cs), 1)
    x16 = [ops.to_s16(dev(x)) for x in xs]
    got = ops.conv2d_multi(pk, x16, act=ops.ACT_RELU, epi=ops.EPI_RES_RELU, e0=dev(res))
    assert rel_err(got.cpu(), torch.relu(res + torch.relu(conv))) <= tol(512 * 9)
    got = ops.conv2d_multi(pk, x16, epi=ops.EPI_ACCUM, e0=dev(res))
    assert rel_err(got.cpu(), res + conv) <= tol(512 * 9)


@pytest.mark.parametrize("lay", [None, 0, 1, 2, 3])
def test_residual_operand_kept_pre_split(ops, lay, no_ksplit):
    """accflow_conv_desc.e0_fmt: the residual of relu(e0 + relu(conv)) read from an S16 tensor equals the fp32-residual
    result computed from that tensor's value (hi + lo) / 2^4, bit for bit, in every wave layout; S16-only output."""
    g = gen(17)
    B, C, H, W = 2, 96, 24, 64
    x = torch.randn(B, C, H, W, generator=g)
    res = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) * 0.04
    b = torch.randn(C, generator=g) * 0.1
    pk = ops.PackedMulti.from_cat(dev(w), dev(b), [C], 1)
    x16, r16 = ops.to_s16(dev(x)), ops.to_s16(dev(res))
    want = ops.conv2d_multi(pk, [x16], act=ops.ACT_RELU, epi=ops.EPI_RES_RELU, e0=r16.to_float(), lay=lay)
    o16 = ops.S16.empty(B, C, H, W, want.device)
    got = ops.conv2d_multi(pk, [x16], act=ops.ACT_RELU, epi=ops.EPI_RES_RELU, e0=r16, out16=o16, fp32_out=False, lay=lay)
    assert got is o16 and torch.equal(o16.data, ops.to_s16(want).data)
    ref = torch.relu(res + torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1).float()))
    assert rel_err(want.cpu(), ref) <= 1e-5
    with pytest.raises(RuntimeError):       # an S16 residual exists for the residual epilogue only
        ops.conv2d_multi(pk, [x16], epi=ops.EPI_ACCUM, e0=r16)


def test_instance_norm_with_pre_split_residual(ops):
    g = gen(18)
    B, C, H, W = 2, 64, 16, 64
    x = torch.randn(B, C, H, W, generator=g)
    res = torch.randn(B, C, H, W, generator=g).abs()
    w = torch.randn(C, C, 3, 3, generator=g) * 0.05
    pk = ops.PackedMulti.from_cat(dev(w), None, [C], 1)
    y, st = ops.conv2d_multi(pk, [ops.to_s16(dev(x))], want_stats=True)
    r16 = ops.to_s16(dev(res))
    o16 = ops.S16.empty(B, C, H, W, y.device)
    ops.instance_norm(y.clone(), 2, res=r16, stats=st, out16=o16, fp32_out=False)
    want = ops.instance_norm(y.clone(), 2, res=r16.to_float(), stats=st)
    assert torch.equal(o16.data, ops.to_s16(want).data)
    ref = torch.relu(res + torch.relu(F.instance_norm(F.conv2d(x.double(), w.double(), padding=1), eps=1e-5).float()))
    assert float((want.cpu() - ref).abs().max()) <= 2e-5


@pytest.mark.parametrize("shape", [(2, 40, 72), (1, 37, 53), (1, 128, 256)])
def test_stem_kernel_vs_cpu_conv(ops, shape):
    """csrc/conv_stem.hip: the encoders' 7x7 stride-2 stem (extractor.py:140,201-205) with a pre-split ReLU output (cnet /
    context), with a raw fp32 output + InstanceNorm statistics (fnet), on even, odd and C1-size images."""
    B, H, W = shape
    g = gen(19)
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.08
    b = torch.randn(64, generator=g) * 0.1
    y = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=3).float()
    pk = ops.PackedConv(dev(w), dev(b), stride=2, padding=3)
    OH, OW = y.shape[2:]
    o16 = ops.S16.empty(B, 64, OH, OW, "cuda")
    r = ops.conv2d(pk, dev(x), act=ops.ACT_RELU, out16=o16, fp32_out=False)
    assert r is o16
    assert rel_err(o16.to_float().cpu(), torch.relu(y)) <= 5e-6
    raw, st = ops.conv2d(pk, dev(x), want_stats=True)
    assert st is not None and rel_err(raw.cpu(), y) <= 5e-6
    ops.instance_norm(raw, 1, stats=st)
    assert float((raw.cpu() - torch.relu(F.instance_norm(y.double(), eps=1e-5)).float()).abs().max()) <= 2e-5
    assert not ops.guard_tripped()
    ops.conv2d(pk, dev(x * 1.0e4), act=ops.ACT_RELU, out16=o16, fp32_out=False)     # 1e4 * 2^4 leaves fp16's range
    assert ops.guard_tripped()


def test_rejects_bad_descriptors(ops):
    g = gen(16)
    w = dev(torch.randn(64, 32, 3, 3, generator=g))
    pk = ops.PackedMulti.from_cat(w, None, [16, 16], 1)
    a = ops.to_s16(dev(torch.randn(1, 16, 8, 32, generator=g)))
    with pytest.raises(RuntimeError):
        ops.conv2d_multi(pk, [a])                       # one source for a two-source pack
    with pytest.raises(RuntimeError):
        ops.conv2d_multi(pk, [a, ops.to_s16(dev(torch.randn(1, 24, 8, 32, generator=g)))])   # wrong channel count
    with pytest.raises(RuntimeError):
        ops.PackedMulti.from_cat(w, None, [16, 8], 1)   # splits do not add up


def test_partial_sums_of_a_concatenation_conv(ops):
    """conv(cat[a, b, c]) = relu(e0 + conv_b(b) + bias) with e0 = conv_{a,c}([a, c]) without bias (PackCache.multi(in_ranges=...),
    ACCFLOW_EPI_RES_RELU with ACCFLOW_ACT_NONE writing S16): a convolution over a concatenation as the sum of its members' convolutions."""
    from accflow_amd.networks._packs import PackCache
    g = gen(77)
    C, B, H, W = 128, 3, 20, 36
    conv = torch.nn.Conv2d(2 * C + 1, 256, 3, 1, 1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / (9 * (2 * C + 1)) ** 0.5)
        conv.bias.copy_(torch.randn(256, generator=g))
    a, b, c = (torch.randn(B, n, H, W, generator=g) for n in (C, C, 1))
    with torch.no_grad():
        lin = F.conv2d(torch.cat([a, b, c], 1).double(), conv.weight.double(), conv.bias.double(), padding=1).float()
    want = F.relu(lin)
    rms = lin.pow(2).mean(dim=(0, 2, 3), keepdim=True).sqrt()    # of the pre-activation: a channel the ReLU nearly empties has no RMS of its own
    conv = conv.cuda()
    pk = PackCache()
    a16, b16, c16 = (ops.to_s16(dev(t)) for t in (a, b, c))
    e0 = ops.conv2d_multi(pk.multi("pre", conv, in_ranges=[(0, C), (2 * C, 2 * C + 1)], with_bias=False), [a16, c16])
    for bsl in ((0, B), (1, 2)):      # the whole batch, and one step's rows of a batched partial sum (B = 1: split-K + reduce)
        out16 = ops.S16.empty(bsl[1] - bsl[0], 256, H, W, a16.device)
        ops.conv2d_multi(pk.multi("f", conv, in_ranges=[(C, 2 * C)]), [b16.batch(*bsl)], epi=ops.EPI_RES_RELU, e0=e0[bsl[0]:bsl[1]],
                         out16=out16, fp32_out=False)
        err = float(((out16.to_float().cpu() - want[bsl[0]:bsl[1]]).abs() / rms).max())
        assert err < tol(9 * (2 * C + 1)), (bsl, err)
