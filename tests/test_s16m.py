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


PROJ = [
    # Cin, planes, B, H, W
    (64, 96, 2, 48, 96),          # layer2 block 0 (extractor.py:9,52-53): the 96-channel layout
    (64, 96, 1, 35, 61),          # odd sizes, one item
    (96, 128, 3, 30, 64),         # layer3 block 0: the 128-channel layout
    (96, 128, 1, 17, 41),
    (32, 64, 2, 20, 70),          # a 64-channel block
]


@pytest.mark.parametrize("case", PROJ)
def test_strided_conv_with_its_projection_in_one_launch(ops, case, no_ksplit):
    """accflow_conv_desc.split_c0 (round 6): a residual block's stride-2 3x3 convolution and the 1x1 stride-2 projection of
    the same input (extractor.py:9,52-53) as ONE launch - output channels [0, planes) = relu(conv1(x)), [planes, 2 planes)
    = the projection, no activation - against float64 F.conv2d of each, pre-split and fp32 destinations, with BatchNorm-
    style row scales, and bit for bit against the two separate launches it replaces (same products, same order)."""
    Cin, planes, B, H, W = case
    g = gen(21)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(planes, Cin, 3, 3, generator=g) * (1.0 / (Cin * 9)) ** 0.5
    b = torch.randn(planes, generator=g) * 0.1
    wp = torch.randn(planes, Cin, 1, 1, generator=g) * (1.0 / Cin) ** 0.5
    bp = torch.randn(planes, generator=g) * 0.1
    sc, scp = torch.rand(planes, generator=g) + 0.5, torch.rand(planes, generator=g) * 4.0 + 0.1
    lin1 = F.conv2d(x.double(), w.double() * sc.double().view(-1, 1, 1, 1), b.double(), stride=2, padding=1).float()
    lin2 = F.conv2d(x.double(), wp.double() * scp.double().view(-1, 1, 1, 1), bp.double(), stride=2).float()
    OH, OW = lin1.shape[2:]
    assert tuple(lin2.shape[2:]) == (OH, OW)
    pk = ops.PackedMulti.from_strided_with_projection(dev(w), dev(b), 1, dev(wp), dev(bp), scale=dev(sc), scale_proj=dev(scp))
    assert pk.split_c0 == planes and pk.Cout == 2 * planes
    x16 = ops.to_s16(dev(x))
    both16 = ops.S16.empty(B, 2 * planes, OH, OW, x16.device)
    both = ops.conv2d_multi(pk, [x16] * len(pk.C), act=ops.ACT_RELU, out16=both16, out_hw=(OH, OW))
    got = both.cpu()
    # errors relative to the RMS of the PRE-activation (a channel the ReLU nearly empties has no RMS of its own)
    rms1 = lin1.pow(2).mean(dim=(0, 2, 3), keepdim=True).sqrt()
    rms2 = lin2.pow(2).mean(dim=(0, 2, 3), keepdim=True).sqrt()
    assert float(((got[:, :planes] - F.relu(lin1)).abs() / rms1).max()) <= 5e-6
    assert float(((got[:, planes:] - lin2).abs() / rms2).max()) <= 5e-6          # (negative values survive: no activation)
    assert float(lin2.min()) < -0.1
    assert torch.equal(both16.to_float().cpu(), got) or float((both16.to_float().cpu() - got).abs().max()) <= 2e-6 * float(got.abs().max())
    # the two launches it replaces
    a = ops.conv2d_multi(ops.PackedMulti.from_strided(dev(w), dev(b), 1, scale=dev(sc)), [x16] * 4, act=ops.ACT_RELU, out_hw=(OH, OW))
    p_ = ops.conv2d_multi(ops.PackedMulti.from_strided(dev(wp), dev(bp), 0, scale=dev(scp)), [x16], out_hw=(OH, OW))
    assert torch.equal(both[:, :planes], a) and torch.equal(both[:, planes:], p_)
    # raw outputs + InstanceNorm statistics of both halves (the feature encoder's form)
    raw, st = ops.conv2d_multi(pk, [x16] * len(pk.C), want_stats=True, out_hw=(OH, OW))
    assert st is not None and tuple(st.partial.shape[:2]) == (B, 2 * planes)
    want = torch.cat([lin1, lin2], dim=1)
    assert float(((raw.cpu() - want).abs() / torch.cat([rms1, rms2], dim=1)).max()) <= 5e-6
    normed = ops.instance_norm(raw.clone(), 0, stats=st).cpu()
    assert float((normed - F.instance_norm(want.double(), eps=1e-5).float()).abs().max()) <= 3e-5
    assert not ops.guard_tripped()
    # descriptors the form does not cover are refused, not mis-computed
    with pytest.raises(RuntimeError):
        ops.conv2d_multi(pk, [x16] * len(pk.C), act=ops.ACT_SIGMOID, out_hw=(OH, OW))
    with pytest.raises(RuntimeError):
        ops.conv2d_multi(pk, [x16] * len(pk.C), epi=ops.EPI_RES_RELU, act=ops.ACT_RELU, e0=both, out_hw=(OH, OW))


def test_chain_producers_write_pre_split(ops):
    """Round 6: the fusion chain's occlusion / error maps (AccFlow_.py:127-135), blended features (:122-124) and deformable
    columns (:102-104) leave their kernels PRE-SPLIT (accflow_get_occ_s16 / accflow_blend_s16 / accflow_deform_columns_s16,
    the latter applying the modulation's sigmoid itself) - each against the oracle's fp32 result, to the S16 format's 22 bits,
    incl. ragged sizes, out-of-image flows / offsets and the thresholded map's both classes."""
    from oracle import accflow_oracle as O
    g = gen(31)
    B, C, H, W = 2, 37, 20, 28
    img, img2 = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    flow = torch.randn(B, 2, H, W, generator=g) * 4
    flow[0, :, 0, :3] = torch.tensor([[-40.0, 0.0, 27.0], [0.0, -1.0, 19.0]])
    e16 = ops.get_occ(dev(flow), dev(img), dev(img2), binary=False, out16=ops.S16.empty(B, C, H, W, "cuda"))
    want = O.get_occ(flow, img, img2, binary=False)
    assert float((e16.to_float().cpu() - want).abs().max()) <= 1e-5
    err = O.get_occ_error(flow, img, img2)
    sc = 1.0 / float(err.median())              # threshold at the median: both classes present
    o16 = ops.get_occ(dev(flow), dev(img * sc), dev(img2 * sc), binary=True, out16=ops.S16.empty(B, 1, H, W, "cuda"))
    ob = o16.to_float().cpu()
    assert torch.equal(ob, ops.get_occ(dev(flow), dev(img * sc), dev(img2 * sc), binary=True).cpu())    # the fp32 kernel's bits
    ref, err = O.get_occ(flow, img * sc, img2 * sc), O.get_occ_error(flow, img * sc, img2 * sc)
    assert 0.2 < float(ref.mean()) < 0.8
    flips = ob != ref
    assert not bool(flips.any()) or bool(((err[flips] - 1.0).abs() < 1e-5).all())
    assert set(ob.unique().tolist()) <= {0.0, 1.0}
    # a large map takes the one-thread-per-pixel kernel: same bits as the fp32 form there too
    big = [torch.randn(3, 24, 240, 200, generator=g) for _ in range(2)]
    fl = torch.randn(3, 2, 240, 200, generator=g) * 3
    ob = ops.get_occ(dev(fl), dev(big[0] * 0.9), dev(big[1] * 0.9), binary=True, out16=ops.S16.empty(3, 1, 240, 200, "cuda"))
    assert torch.equal(ob.to_float(), ops.get_occ(dev(fl), dev(big[0] * 0.9), dev(big[1] * 0.9), binary=True))
    # blend
    f1, f2 = torch.randn(B, 24, H, W, generator=g) * 3, torch.randn(B, 24, H, W, generator=g)
    m = torch.rand(B, 1, H, W, generator=g)
    b16 = ops.blend(dev(f1), dev(f2), dev(m), out16=ops.S16.empty(B, 24, H, W, "cuda"))
    want = f1 * m + (1 - m) * f2
    assert float((b16.to_float().cpu() - want).abs().max()) <= 2e-6 * float(want.abs().max())
    # deformable convolution from pre-split columns with the sigmoid inside the columns kernel
    C2 = 128
    x = torch.randn(B, C2, 14, 22, generator=g)
    off = torch.randn(B, 18, 14, 22, generator=g) * 2.5
    off[0, :, :2] *= 6
    logit = torch.randn(B, 9, 14, 22, generator=g) * 2
    w = torch.randn(C2, C2, 3, 3, generator=g) * 0.04
    bias = torch.randn(C2, generator=g) * 0.1
    ref = O.deform_conv2d(x, off, torch.sigmoid(logit), w, bias)
    pk = ops.PackedConv(dev(w), dev(bias), stride=1, padding=1, tap_major=True)
    om = dev(torch.cat([off, logit], 1)).contiguous()          # (the channel slices AccPlus passes)
    out16 = ops.S16.empty(B, C2, 14, 22, "cuda")
    ops.deform_conv2d_s16(pk, dev(x), om[:, :18], om[:, 18:], out16, mask_is_logit=True)
    rms = float(ref.pow(2).mean().sqrt())
    assert float((out16.to_float().cpu() - ref).abs().max()) <= 1e-4 * max(1.0, rms)
    out16b = ops.S16.empty(B, C2, 14, 22, "cuda")
    ops.deform_conv2d_s16(pk, dev(x), dev(off), dev(torch.sigmoid(logit)), out16b)      # (mask given, no sigmoid inside)
    assert float((out16b.to_float() - out16.to_float()).abs().max()) <= 2e-5 * max(1.0, rms)
    assert not ops.guard_tripped()


def test_fnet_block_with_projection_vs_oracle(ops):
    """The InstanceNorm encoder's projected residual block (extractor.py:51-63, layer2 / layer3 block 0) on the round-6 path:
    conv1 + projection in one launch (raw + statistics), conv2 normalising on load, the closing pass normalising the
    projection itself (accflow_instance_norm_apply_s16proj_f32) - against the float64 formula and against the round-5 path
    (separate launches, ACCFLOW_FUSE_PROJECTION=0)."""
    from accflow_amd.networks.raft import extractor as E
    from accflow_amd.networks._packs import PackCache
    g = gen(41)
    for cin, planes, B, H, W in ((64, 96, 2, 40, 72), (96, 128, 1, 30, 64)):
        blk = E.ResidualBlock(cin, planes, "instance", stride=2)
        with torch.no_grad():
            for prm in blk.parameters():
                prm.copy_(torch.randn(prm.shape, generator=g) * (0.1 if prm.dim() == 1 else (1.0 / prm[0].numel()) ** 0.5))
        x = torch.relu(torch.randn(B, cin, H, W, generator=g))
        with torch.no_grad():
            d = lambda t: t.double()  # noqa: E731
            y = torch.relu(F.instance_norm(F.conv2d(d(x), d(blk.conv1.weight), d(blk.conv1.bias), stride=2, padding=1), eps=1e-5))
            y = torch.relu(F.instance_norm(F.conv2d(y, d(blk.conv2.weight), d(blk.conv2.bias), padding=1), eps=1e-5))
            r = F.instance_norm(F.conv2d(d(x), d(blk.downsample[0].weight), d(blk.downsample[0].bias), stride=2), eps=1e-5)
            want = torch.relu(r + y).float()
        blk = blk.cuda()
        x16 = ops.to_s16(dev(x))
        got = blk.run16(x16, PackCache(), "t").to_float().cpu()
        assert float((got - want).abs().max()) <= 5e-5, float((got - want).abs().max())
        old = E.FUSE_PROJECTION
        try:
            E.FUSE_PROJECTION = False
            ref5 = blk.run16(x16, PackCache(), "t").to_float().cpu()
        finally:
            E.FUSE_PROJECTION = old
        assert float((got - ref5).abs().max()) <= 2e-5
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
