"""The pre-split "S16" activation format (include/accflow_hip.h, accflow_conv_desc.in_fmt / out16): producers write the
fp16 hi / lo terms the matrix-core kernels multiply, consumers stage them by LDS DMA.  The split is the same function
of the same fp32 value wherever it is performed, so EVERYTHING here is checked for bit-identity against the fp32-
activation path of the same f16x3 arithmetic (which tests/test_hip_parity.py in turn checks against the oracle and the
reference's fixtures)."""
import pytest
import torch

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


def test_to_s16_matches_the_split_and_round_trips(ops):
    g = gen(1)
    x = torch.randn(3, 21, 9, 13, generator=g) * torch.tensor([1e-3, 1.0, 50.0]).view(3, 1, 1, 1)
    a = ops.to_s16(dev(x))
    b = ops.S16.from_float(dev(x))
    assert torch.equal(a.data, b.data)                                 # kernel split == torch restatement of the split
    assert a.shape == (3, 21, 9, 13) and a.data.shape[1] == 3          # 21 channels -> 3 octets, tail channels zero
    back = a.to_float().cpu()
    assert float((back - x).abs().max()) <= 2.0 ** -21 * float(x.abs().max())   # hi + lo carries >= 22 bits
    sl = a.channels(8, 21)
    assert torch.equal(sl.to_float().cpu(), back[:, 8:21]) and sl.bs == a.bs
    assert not ops.guard_tripped()
    ops.to_s16(dev(torch.full((1, 8, 4, 4), 5000.0)))                  # 5000 * 2^4 does not fit fp16
    assert ops.guard_tripped()


CASES = [
    # Cin(s), Cout, KH, KW, B, H, W
    ((128,), 256, 3, 3, 2, 20, 40),
    ((256,), 192, 3, 3, 1, 12, 64),        # 128 + 64 channel cut of the launch
    ((128,), 64, 3, 3, 2, 13, 37),         # 64-channel kernel, ragged tiles
    # (1x1 convolutions of < 512 channels on small grids go to the im2col kernel in the fp32-activation path - another
    # summation order - so the bit comparison needs grids of >= 300 workgroups, where both paths run the direct kernel)
    ((352,), 256, 1, 1, 3, 60, 128),       # 1x1: 32-channel chunks
    ((324,), 256, 1, 1, 2, 57, 131),       # 1x1 with a channel tail (41 octets) and ragged tiles
    ((128, 128), 256, 1, 5, 2, 12, 48),    # two sources
    ((128, 256), 128, 5, 1, 1, 20, 33),
    ((16,), 128, 1, 7, 2, 16, 32),         # one chunk
    ((256,), 126, 3, 3, 2, 12, 32),        # partial last octet of the S16 output
    ((64,), 96, 3, 3, 1, 24, 64),
    ((256,), 576, 1, 1, 2, 60, 128),
    ((256,), 27, 3, 3, 1, 12, 32),         # odd channel count: the partner of the last channel is written as zero
]


@pytest.mark.parametrize("case", CASES)
def test_conv_s16_in_and_out_bit_identical(ops, case):
    cins, Cout, KH, KW, B, H, W = case
    g = gen(7)
    xs = [torch.randn(B, c, H, W, generator=g) for c in cins]
    Cin = sum(cins)
    w = torch.randn(Cout, Cin, KH, KW, generator=g) * (1.0 / (Cin * KH * KW)) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    pk = ops.PackedConv(dev(w), dev(b), padding=(KH // 2, KW // 2), C0=cins[0])
    x32 = [dev(x) for x in xs]
    x16 = [ops.to_s16(x) for x in x32]
    for act in (ops.ACT_NONE, ops.ACT_RELU):
        want = ops.conv2d(pk, x32[0], x32[1] if len(xs) > 1 else None, act=act)
        # S16 in, fp32 out
        got = ops.conv2d(pk, x16[0], x16[1] if len(xs) > 1 else None, act=act)
        assert torch.equal(got, want), "S16 input changes the result"
        # fp32 in, fp32 + S16 out; the S16 copy must be exactly the split of the fp32 result
        o16 = ops.S16.empty(B, Cout, H, W, want.device)
        o16.data.fill_(0x7B7B7B7B)                                     # sentinel: shows what the epilogue leaves alone
        got2 = ops.conv2d(pk, x32[0], x32[1] if len(xs) > 1 else None, act=act, out16=o16)
        assert torch.equal(got2, want)
        ref16 = ops.to_s16(want)
        O = (Cout + 7) // 8
        halfs_got = o16.data.view(torch.int16).view(B, O, 2, H, W, 8)
        halfs_ref = ref16.data.view(torch.int16).view(B, O, 2, H, W, 8)
        nw = (Cout + 1) // 2 * 2                                       # channel pairs written (odd tail: partner = 0)
        for o in range(O):
            k = min(8, nw - 8 * o)
            assert torch.equal(halfs_got[:, o, :, :, :, :k], halfs_ref[:, o, :, :, :, :k]), (o, k)
            if k < 8:                                                  # the rest of a partial octet belongs to the caller
                assert bool((halfs_got[:, o, :, :, :, k:] == 0x7B7B).all())
        # S16 in, S16 out only
        o16b = ops.S16.empty(B, Cout, H, W, want.device)
        o16b.data.fill_(0x7B7B7B7B)
        r = ops.conv2d(pk, x16[0], x16[1] if len(xs) > 1 else None, act=act, out16=o16b, fp32_out=False)
        assert r is o16b and torch.equal(o16b.data, o16.data)
    assert not ops.guard_tripped()


def test_conv_s16_gru_epilogues_bit_identical(ops):
    """GRU_ZR (z fp32, r*h pre-split only) and GRU_Q (h fp32 + pre-split) with the hoisted-context addend, and the
    residual / accumulate epilogues."""
    g = gen(9)
    B, H, W, hd = 2, 16, 48, 128
    h = torch.randn(B, hd, H, W, generator=g) * 0.5
    x = torch.randn(B, 128, H, W, generator=g)
    wzr = torch.randn(2 * hd, 256, 1, 5, generator=g) * 0.03
    wq = torch.randn(hd, 256, 1, 5, generator=g) * 0.03
    pzr = ops.PackedConv(dev(wzr), dev(torch.randn(2 * hd, generator=g) * 0.1), padding=(0, 2), C0=hd)
    pq = ops.PackedConv(dev(wq), dev(torch.randn(hd, generator=g) * 0.1), padding=(0, 2), C0=hd)
    pre_zr, pre_q = dev(torch.randn(B, 2 * hd, H, W, generator=g) * 0.2), dev(torch.randn(B, hd, H, W, generator=g) * 0.2)
    h32, x32 = dev(h), dev(x)
    z = torch.empty_like(h32); rh = torch.empty_like(h32)
    ops.conv2d(pzr, h32, in1=x32, out=z, act=ops.ACT_SIGMOID, epi=ops.EPI_GRU_ZR, e0=h32, out2=rh, pre=pre_zr)
    hn = ops.conv2d(pq, rh, in1=x32, act=ops.ACT_TANH, epi=ops.EPI_GRU_Q, e0=h32, e1=z, pre=pre_q)
    h16, x16 = ops.to_s16(h32), ops.to_s16(x32)
    z2 = torch.empty_like(h32)
    rh16 = ops.S16.empty(B, hd, H, W, h32.device)
    ops.conv2d(pzr, h16, in1=x16, out=z2, act=ops.ACT_SIGMOID, epi=ops.EPI_GRU_ZR, e0=h32, out16=rh16, fp32_out=False, pre=pre_zr)
    assert torch.equal(z2, z) and torch.equal(rh16.data, ops.to_s16(rh).data)
    hstate = h32.clone()
    hn16 = ops.S16.empty(B, hd, H, W, h32.device)
    ops.conv2d(pq, rh16, in1=x16, out=hstate, act=ops.ACT_TANH, epi=ops.EPI_GRU_Q, e0=hstate, e1=z2, out16=hn16, pre=pre_q)
    assert torch.equal(hstate, hn) and torch.equal(hn16.data, ops.to_s16(hn).data)      # in place on the fp32 state
    # residual + relu, accumulate
    w3 = torch.randn(128, 128, 3, 3, generator=g) * 0.03
    p3 = ops.PackedConv(dev(w3), dev(torch.zeros(128)), padding=1)
    want = ops.conv2d(p3, x32, act=ops.ACT_RELU, epi=ops.EPI_RES_RELU, e0=h32)
    o16 = ops.S16.empty(B, 128, H, W, h32.device)
    got = ops.conv2d(p3, x16, act=ops.ACT_RELU, epi=ops.EPI_RES_RELU, e0=h32, out16=o16)
    assert torch.equal(got, want) and torch.equal(o16.data, ops.to_s16(want).data)
    # <= 4 output channels: all taps as one 1x1 conv over the S16 input + the tap sum (flow head), at the working size
    # (on small grids the fp32-activation path picks other kernels - another summation order)
    B, H, W = 6, 60, 128
    xb = dev(torch.randn(B, 256, H, W, generator=g))
    w2 = torch.randn(2, 256, 3, 3, generator=g) * 0.05
    p2 = ops.PackedConv(dev(w2), dev(torch.randn(2, generator=g)), padding=1)
    co = dev(torch.randn(B, 2, H, W, generator=g))
    c1 = co.clone()
    want = ops.conv2d(p2, xb, out=c1, epi=ops.EPI_ACCUM, e0=c1)
    c2 = co.clone()
    got = ops.conv2d(p2, ops.to_s16(xb), out=c2, epi=ops.EPI_ACCUM, e0=c2)
    assert torch.equal(got, want)


@pytest.mark.parametrize("hw", [(16, 48), (13, 37), (60, 128)])
def test_gru_epilogues_on_packed_operands(ops, hw, monkeypatch):
    """Round 6: inside the refinement loop the GRU epilogues read the state h from its PRE-SPLIT tensor (accflow_conv_desc.e0_fmt)
    and z / the context addend from PIXEL-MAJOR fp32 tensors (accflow_conv_desc.p32), z is written pixel-major and the q launch
    writes no fp32 state.  Same arithmetic as the fp32-operand form of the same convolutions: bit-identical when the fp32 state
    equals the pre-split one's value (h = (hi + lo) / 2^4 exactly), for both kernel shapes (1x5, 5x1), ragged sizes, and
    in place on the state tensor; the context convolutions' pixel-major store is a permutation of the plain one."""
    from accflow_amd.networks.raft.update import h16_state_supported
    if not h16_state_supported():
        pytest.skip("the packed-operand GRU epilogue rides on the tap-specialised 5-tap kernels")
    # (small grids: the fp32-operand form would split K - another summation order; the packed form never does)
    monkeypatch.setattr(ops, "USE_KSPLIT", False)
    H, W = hw
    g = gen(19)
    B, hd = 2, 128
    h16 = ops.to_s16(dev(torch.randn(B, hd, H, W, generator=g) * 0.5))
    h32 = h16.to_float()                                   # the fp32 state with exactly the pre-split tensor's value
    x16 = ops.to_s16(dev(torch.randn(B, 128, H, W, generator=g)))
    for kh, kw in ((1, 5), (5, 1)):
        pad = (kh // 2, kw // 2)
        pzr = ops.PackedConv(dev(torch.randn(2 * hd, 256, kh, kw, generator=g) * 0.03), dev(torch.randn(2 * hd, generator=g) * 0.1),
                             padding=pad, C0=hd)
        pq = ops.PackedConv(dev(torch.randn(hd, 256, kh, kw, generator=g) * 0.03), dev(torch.randn(hd, generator=g) * 0.1),
                            padding=pad, C0=hd)
        pre_zr, pre_q = dev(torch.randn(B, 2 * hd, H, W, generator=g) * 0.2), dev(torch.randn(B, hd, H, W, generator=g) * 0.2)
        # the fp32-operand form (rounds 3-5)
        z = torch.empty_like(h32)
        rh16 = ops.S16.empty(B, hd, H, W, h32.device)
        ops.conv2d(pzr, h16, in1=x16, out=z, act=ops.ACT_SIGMOID, epi=ops.EPI_GRU_ZR, e0=h32, out16=rh16, fp32_out=False, pre=pre_zr)
        hn32 = h32.clone()
        hn16 = ops.S16.empty(B, hd, H, W, h32.device)
        ops.conv2d(pq, rh16, in1=x16, out=hn32, act=ops.ACT_TANH, epi=ops.EPI_GRU_Q, e0=hn32, e1=z, out16=hn16, pre=pre_q)
        # the packed-operand form
        zp = torch.empty_like(h32)
        rh16p = ops.S16.empty(B, hd, H, W, h32.device)
        ops.conv2d(pzr, h16, in1=x16, out=zp, act=ops.ACT_SIGMOID, epi=ops.EPI_GRU_ZR, e0=h16, out16=rh16p, fp32_out=False,
                   pre=ops.to_p32(pre_zr))
        assert torch.equal(ops.from_p32(zp), z) and torch.equal(rh16p.data, rh16.data)
        state = ops.S16(h16.data.clone(), hd)              # in place: e0 and out16 are the same tensor
        ops.conv2d(pq, rh16p, in1=x16, out=None, act=ops.ACT_TANH, epi=ops.EPI_GRU_Q, e0=state, e1=zp, out16=state, fp32_out=False,
                   pre=ops.to_p32(pre_q))
        assert torch.equal(state.data, hn16.data)
    pc = ops.PackedConv(dev(torch.randn(256, 128, 1, 5, generator=g) * 0.05), None, padding=(0, 2))
    plain = ops.conv2d(pc, x16)
    assert torch.equal(ops.from_p32(ops.conv2d(pc, x16, p32_out=True)), plain)
    assert torch.equal(ops.from_p32(ops.to_p32(plain)), plain)
    with pytest.raises(RuntimeError):     # not a multiple of 128 output channels
        ops.conv2d(ops.PackedConv(dev(torch.randn(64, 128, 1, 5, generator=g)), None, padding=(0, 2)), x16, p32_out=True)
    with pytest.raises(RuntimeError):     # a pre-split e0 belongs to the GRU epilogues
        ops.conv2d(pc, x16, epi=ops.EPI_RES_RELU, act=ops.ACT_RELU, e0=ops.S16.empty(B, 256, H, W, h32.device))
    assert not ops.guard_tripped()


@pytest.mark.parametrize("shape", [(2, 16, 32), (1, 17, 24), (3, 60, 128)])
def test_lookup_and_flow_s16(ops, shape):
    B, H8, W8 = shape
    g = gen(11)
    f1 = dev(torch.randn(B, 256, H8, W8, generator=g))
    f2 = dev(torch.randn(B, 256, H8, W8, generator=g))
    pyr = ops.corr_volume_disp(f1, f2)
    coords = ops.coords_grid(B, H8, W8, f1.device) + dev(torch.randn(B, 2, H8, W8, generator=g) * 3.0)
    want = ops.corr_lookup(pyr, coords)                                         # (B, 324, h, w), channel l*81 + i*9 + j
    c88 = torch.zeros((B, 4, 88, H8, W8), device=f1.device)
    c88[:, :, :81] = want.view(B, 4, 9, 9, H8, W8).transpose(2, 3).reshape(B, 4, 81, H8, W8)
    out16 = ops.S16.empty(B, ops.LOOKUP_S16_CHANNELS, H8, W8, f1.device)
    ops.corr_lookup_s16(pyr, coords, out16)
    assert torch.equal(out16.data, ops.to_s16(c88.view(B, 352, H8, W8)).data)
    # flow pieces
    d0 = torch.empty((B, 2, H8, W8), device=f1.device); d1 = torch.empty_like(d0)
    st = torch.empty((B, 16, H8, W8), device=f1.device)
    ops.flow_from_coords(coords, dst0=d0, dst1=d1, stack16=st)
    e0 = torch.empty_like(d0); e1 = torch.empty_like(d0)
    st16 = ops.S16.empty(B, 16, H8, W8, f1.device)
    m16 = ops.S16.empty(B, 128, H8, W8, f1.device)
    m16.data.fill_(0x7B7B7B7B)
    ops.flow_from_coords_s16(coords, e0, e1, st16, m16, 126)
    assert torch.equal(e0, d0) and torch.equal(e1, d1) and torch.equal(st16.data, ops.to_s16(st).data)
    halfs = m16.data.view(torch.int16).view(B, 16, 2, H8, W8, 8)
    ref = ops.to_s16(d0).data.view(torch.int16).view(B, 1, 2, H8, W8, 8)
    assert torch.equal(halfs[:, 15, :, :, :, 6:8], ref[:, 0, :, :, :, 0:2])
    assert bool((halfs[:, 15, :, :, :, :6] == 0x7B7B).all()) and bool((halfs[:, :15] == 0x7B7B).all())


@pytest.mark.parametrize("name", ["raft", "gma"])
def test_estimator_s16_on_off(ops, name):
    """The whole estimator with the S16 iteration (lookup -> motion encoder -> GRU -> heads on pre-split tensors) against
    the fp32-activation iteration: equal to fp32 rounding (every convolution but convc1 is bit-identical on its own -
    tests above - when both paths pick the same kernel; small grids divert the fp32-activation path to other kernels)."""
    from oracle import accflow_oracle as O
    from accflow_amd.data.synthetic import make_sequence, make_state_dict, normalize
    from accflow_amd.networks import build_flow_estimator
    m = build_flow_estimator(name)
    m.load_state_dict(make_state_dict(m), strict=True)
    m = m.cuda().eval()
    saved = ops.USE_S16

    def both(fn):
        res = {}
        try:
            for flag in (True, False):
                ops.USE_S16 = flag
                res[flag] = fn()
        finally:
            ops.USE_S16 = saved
        return res[True], res[False]

    fr = [dev(normalize(f)) for f in make_sequence(1003, 3, 128, 256, batch=2)]
    i1, i2 = torch.cat([fr[1], fr[2]]), torch.cat([fr[0], fr[0]])              # 4 items: two pair-group streams
    init = dev(torch.randn(4, 2, 16, 32, generator=gen(5)))
    for a, b in zip(*both(lambda: (m(i1, i2, iters=3), m(i1, i2, iters=2, flow_init=init)))):
        me, mx = O.epe(a.cpu(), b.cpu())
        assert me <= 3e-5 and mx <= 1e-3, (me, mx)      # (two fp32-equivalent paths; the parity gate vs the reference is 1e-3)
    if name == "raft":
        import accflow_amd.networks.raft.raft as R
        big = [dev(normalize(f)) for f in make_sequence(1004, 2, 480, 1024, batch=5)]
        streams, R.N_STREAMS = R.N_STREAMS, 1            # one group of 5 pairs: 300 pixel tiles per launch
        try:
            a, b = both(lambda: m(big[1], big[0], iters=2))
        finally:
            R.N_STREAMS = streams
        # (not bit-equal by construction: the S16 lookup hands convc1 its 324 channels in another order - level-major with
        # the window transposed - so that convolution sums the same products in another order; everything else is)
        me, mx = O.epe(a.cpu(), b.cpu())
        print("480x1024 x5, 2 iterations: S16 vs fp32-activation path EPE mean %.2e max %.2e" % (me, mx))
        assert me <= 1e-5 and mx <= 1e-3, (me, mx)
    # the reference-signature entry of the update block (fp32 tensors in and out) also runs the S16 iteration
    ub = m.update_block
    g = gen(3)
    net, inp = dev(torch.tanh(torch.randn(2, 128, 16, 32, generator=g))), dev(torch.relu(torch.randn(2, 128, 16, 32, generator=g)))
    corr, flow = dev(torch.randn(2, 324, 16, 32, generator=g)), dev(torch.randn(2, 2, 16, 32, generator=g))
    att = m.att(inp) if name == "gma" else None
    ra, rb = both(lambda: ub(net, inp, corr, flow, att) if name == "gma" else ub(net, inp, corr, flow))
    for a, b in zip(ra, rb):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))


# ((1, 36, 40): an odd count of 32-channel chunks over split-K parts and a ragged tile column)
@pytest.mark.parametrize("shape", [(2, 16, 32), (1, 17, 23), (1, 36, 40), (1, 60, 128)])
def test_gma_attention_and_aggregation_s16(ops, shape):
    """The attention built straight into its pre-split form (two passes of a register-only q.k GEMM, no logits matrix) and
    the aggregation GEMM on it with the channel-block scatter, against the fp32 formula."""
    B, h, w = shape
    P, D = h * w, 128
    g = gen(13)
    qk = dev(torch.randn(B, 2 * D, h, w, generator=g) * 1.5)
    a16 = ops.gma_attention_s16(qk, D, D ** -0.5)
    att = a16.to_float().view(B, P, P)                                      # [b][j][i]
    q, k = qk[:, :D].reshape(B, D, P), qk[:, D:].reshape(B, D, P)
    ref = torch.softmax(torch.einsum("bdi,bdj->bij", q.double(), k.double()) * D ** -0.5, dim=2).transpose(1, 2).float()
    assert float((att - ref).abs().max()) <= 2e-6 and float((att.sum(1) - 1).abs().max()) <= 1e-5
    assert not ops.guard_tripped()
    # aggregation: two items share attention 0 (stacked along the rows of one GEMM), residual / outputs in place in slices
    n = 2
    big = dev(torch.randn(n, 3 * D, h, w, generator=g))                     # [fmap | out | unused] slices of one buffer
    fmap, out = big[:, :D], big[:, D:2 * D]
    v = dev(torch.randn(n, D, h, w, generator=g))
    gamma = dev(torch.tensor([0.37]))
    out16 = ops.S16.empty(n, 2 * D, h, w, qk.device)
    ops.gma_aggregate_s16(a16.ptr(), v, fmap[0].data_ptr(), big.stride(0), gamma, out[0].data_ptr(), big.stride(0),
                          out16.channels(D, 2 * D).ptr(), out16.bs, n, D, h, w)
    want = fmap.double() + 0.37 * torch.einsum("ji,ndj->ndi", ref[0].double(), v.reshape(n, D, P).double()).view(n, D, h, w)
    assert float((out.double() - want).abs().max()) <= 3e-5
    assert torch.equal(out16.channels(D, 2 * D).data, ops.to_s16(out.contiguous()).data)


@pytest.mark.parametrize("env", [{"ACCFLOW_CORR_GEMM": "regs"}])
def test_correlation_gemm_variants(env):
    """The selectable register-only operand loop of the displaced correlation GEMM (the default is the LDS ring) passes the
    same exact-permutation / lookup tests - in a fresh process, because the switch is read once."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_hip_parity.py"), "-m", "gpu", "-q", "-x",
                        "-k", "test_corr_disp_volume_and_lookup or test_corr_per_frame_packs or test_full_size_properties"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, **env), cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
