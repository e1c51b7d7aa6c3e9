"""Pin the CPU oracle against the golden vectors produced by the imported reference
(tests/golden/make_golden.py).  CPU only; every oracle function on the path is covered."""
import numpy as np
import pytest
import torch

from accflow_amd.data.synthetic import make_sequence, make_state_dict, normalize
from accflow_amd.networks import build_flow_estimator
from accflow_amd.networks.AccFlow_ import AccFlow
from oracle import accflow_oracle as O

T = torch.from_numpy


def close(a, b, atol, rtol=1e-4, what=""):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    assert (err <= tol).all(), "%s: max err %.3e (tol %.1e) at %s" % (what, err.max(), atol, np.unravel_index(err.argmax(), err.shape))


@pytest.fixture(scope="module")
def raft_sd():
    return make_state_dict(build_flow_estimator("raft"))


@pytest.fixture(scope="module")
def gma_sd():
    return make_state_dict(build_flow_estimator("gma"))


def pair(seed, H, W):
    fr = [normalize(f) for f in make_sequence(seed, 2, H, W)]
    return fr[1], fr[0]


def test_state_dict_inventory():
    """key counts / parameter totals of SURVEY 8(b) (probe dump of the reference)."""
    r, g = build_flow_estimator("raft"), build_flow_estimator("gma")
    a = AccFlow(build_flow_estimator("acc|raft"))
    assert (len(r.state_dict()), sum(p.numel() for p in r.parameters())) == (179, 5257536)
    assert (len(g.state_dict()), sum(p.numel() for p in g.parameters())) == (185, 5879873)
    assert (len(a.state_dict()), sum(p.numel() for p in a.parameters())) == (252, 11757081)
    with pytest.raises(NotImplementedError):
        build_flow_estimator("pwc")


def test_encoders_and_corr(golden, raft_sd):
    g = golden("raft_c1")
    i1, i2 = pair(int(g["seed"]), int(g["H"]), int(g["W"]))
    fm = O.basic_encoder(torch.cat([i1, i2]), raft_sd, "fnet", "instance")
    close(fm[:1], g["fmap1"], 2e-4, what="fmap1")
    close(fm[1:], g["fmap2"], 2e-4, what="fmap2")
    close(O.basic_encoder(i1, raft_sd, "cnet", "batch"), g["cnet"], 2e-4, what="cnet")
    pyr = O.corr_pyramid(T(g["fmap1"]), T(g["fmap2"]))
    sel = g["pyr_sel"]
    for l in range(4):
        close(pyr[l][sel], g["pyr%d" % l], 1e-4, what="pyramid level %d" % l)


def test_lookup(golden):
    g = golden("raft_c1")
    pyr = O.corr_pyramid(T(g["fmap1"]), T(g["fmap2"]))
    B, _, h, w = g["fmap1"].shape
    close(O.corr_lookup(pyr, O.coords_grid(B, h, w)), g["lookup0"], 1e-4, what="lookup at grid")
    close(O.corr_lookup(pyr, T(g["coords_r"])), g["lookup_r"], 1e-4, what="lookup at random coords")


def test_update_block_and_upsample(golden, raft_sd):
    g = golden("raft_c1")
    cnet = T(g["cnet"])
    net, inp = torch.tanh(cnet[:, :128]), torch.relu(cnet[:, 128:])
    B, _, h, w = g["cnet"].shape
    flow = T(g["coords_r"]) - O.coords_grid(B, h, w)
    corr = T(g["lookup_r"])
    close(O.motion_encoder(flow, corr, raft_sd, "update_block.encoder"), g["motion"], 1e-4, what="motion")
    net1, mask1, delta1 = O.update_block(net, inp, corr, flow, raft_sd)
    close(net1, g["ub_net"], 1e-4, what="gru net")
    close(mask1[:, ::9], g["ub_mask_s"], 1e-4, what="mask")
    close(delta1, g["ub_delta"], 1e-4, what="delta")
    close(O.convex_upsample(flow + T(g["ub_delta"]), mask1), g["upsample"], 2e-4, what="convex upsample")


@pytest.mark.parametrize("name", ["raft", "gma"])
def test_estimator_end_to_end(golden, raft_sd, gma_sd, name):
    g = golden(name + "_c1")
    sd = raft_sd if name == "raft" else gma_sd
    i1, i2 = pair(int(g["seed"]), int(g["H"]), int(g["W"]))
    for it in (1, 4, 12):
        out = O.raft_forward(sd, i1, i2, iters=it, gma=(name == "gma"))
        ref = g["flow_it%d" % it]
        out = out if it == 12 else out[:, :, ::2, ::2]
        m, mx = O.epe(out, T(ref))
        assert m < 1e-3 and mx < 1e-2, (name, it, m, mx)
    out = O.raft_forward(sd, i1, i2, iters=4, flow_init=T(g["flow_init"]), gma=(name == "gma"))
    m, mx = O.epe(out[:, :, ::2, ::2], T(g["flow_it4_init"]))
    assert m < 1e-3 and mx < 1e-2, (name, "flow_init", m, mx)


def test_gma_attention_aggregate(golden, gma_sd):
    g = golden("gma_c1")
    cnet = T(g["cnet"])
    inp = torch.relu(cnet[:, 128:])
    attn = O.gma_attention(inp, gma_sd)
    close(attn.sum(-1), g["attn_rowsum"], 1e-5, what="attention row sums")
    close(attn[:, 0, g["attn_rows_sel"]], g["attn_rows"], 1e-6, rtol=1e-3, what="attention rows")
    close(O.gma_aggregate(attn, T(g["motion"]), gma_sd, "update_block.aggregator"), g["motion_global"], 1e-4,
          what="aggregate")


def test_harness_warp_downflow(golden):
    g = golden("harness")
    close(O.backwarp(T(g["img"]), T(g["fflow"])), g["warped"], 1e-5, what="backwarp")
    close(O.downflow8(T(g["big"])), g["down"], 1e-5, what="downflow8")
    occ_bw, occ_fw = O.calc_occ_mask(T(g["bflow"]), T(g["fflow"]))
    assert (occ_bw.numpy() != g["occ_bw"]).mean() < 2e-3 and (occ_fw.numpy() != g["occ_fw"]).mean() < 2e-3
    e = O.cal_epe(T(g["pred"]), T(g["bflow"]), T(g["occ_bw"]))
    for got, key in zip(e, ("epe_all", "epe_occ", "epe_vis")):
        close(got, g[key], 1e-5, what=key)


def test_accflow_step_and_outputs(golden):
    g = golden("accflow_c1")
    model = AccFlow(build_flow_estimator("acc|raft"))
    sd = make_state_dict(model)
    frames = [normalize(f) for f in make_sequence(int(g["seed"]), int(g["n_frames"]), int(g["H"]), int(g["W"]))]
    trace = {}
    outs = O.accflow_forward(sd, frames, trace=trace)
    s2 = trace["step2"]
    for key in ("dflow", "flow_ini", "f_ini", "f", "c1", "f_acc", "f_fuse", "out_small"):
        close(s2[key], g["s2_" + key], 2e-4, rtol=1e-3, what="step2 " + key)
    close(s2["cn"][:, ::4], g["s2_cn"], 2e-4, what="step2 cn")
    flips = (s2["o"].numpy() != g["s2_o"])
    assert not flips.any() or (np.abs(s2["o_err"].numpy()[flips] - 1.0) < 1e-4).all()
    for k, o in enumerate(outs):
        m, mx = O.epe(o, T(g["out%d" % k]))
        assert m < 1e-3 and mx < 1e-2, (k, m, mx)


def test_deform_conv_known_answers():
    """torchvision is absent from the image: identities every deform_conv2d must satisfy (the independent float64
    vectors of make_deform_golden.py are checked in the test below / beside this one)."""
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(2, 6, 9, 11, generator=gen)
    w = torch.randn(5, 6, 3, 3, generator=gen) * 0.2
    b = torch.randn(5, generator=gen)
    zero = torch.zeros(2, 18, 9, 11)
    one = torch.ones(2, 9, 9, 11)
    assert torch.allclose(O.deform_conv2d(x, zero, one, w, b), F.conv2d(x, w, b, padding=1), atol=1e-5)
    assert torch.allclose(O.deform_conv2d(x, zero, 0 * one, w, b), b[None, :, None, None].expand(2, 5, 9, 11), atol=1e-6)
    # integer offsets (dy=+1, dx=-2 on every tap) == conv over the explicitly shifted, zero-padded input
    off = zero.clone()
    off[:, 0::2] = 1.0
    off[:, 1::2] = -2.0
    xs = torch.zeros_like(x)
    xs[:, :, :-1, 2:] = x[:, :, 1:, :-2]
    # shifting the input commutes with the conv only if the conv's own zero padding sees the shifted plane's
    # border, so compare against sampling semantics directly: out(y,x) = sum w * x_zp(y-1+ky+1, x-1+kx-2)
    xp = F.pad(x, (3, 3, 3, 3))
    ref = torch.zeros(2, 5, 9, 11)
    for ky in range(3):
        for kx in range(3):
            patch = xp[:, :, 3 + ky - 1 + 1:3 + ky - 1 + 1 + 9, 3 + kx - 1 - 2:3 + kx - 1 - 2 + 11]
            ref += torch.einsum("oc,nchw->nohw", w[:, :, ky, kx], patch)
    ref += b[None, :, None, None]
    assert torch.allclose(O.deform_conv2d(x, off, one, w, b), ref, atol=1e-5)
    # half-pixel offsets == bilinear grid_sample (interior pixels, where both conventions have all 4 corners)
    off = zero.clone() + 0.5
    got = O.deform_conv2d(x, off, one, w, b)
    ys, xs_ = torch.meshgrid(torch.arange(9.0), torch.arange(11.0), indexing="ij")
    ref = torch.zeros(2, 5, 9, 11)
    for ky in range(3):
        for kx in range(3):
            gx = 2 * (xs_ - 1 + kx + 0.5) / 10 - 1
            gy = 2 * (ys - 1 + ky + 0.5) / 8 - 1
            grid = torch.stack([gx, gy], -1)[None].expand(2, -1, -1, -1)
            samp = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
            ref += torch.einsum("oc,nchw->nohw", w[:, :, ky, kx], samp)
    ref += b[None, :, None, None]
    assert torch.allclose(got[:, :, 1:-2, 1:-2], ref[:, :, 1:-2, 1:-2], atol=1e-5)


def test_deform_conv_vs_independent_known_answers(golden):
    """The oracle's deform_conv2d against tests/golden/deform_conv_kat.npz: float64 scalar-loop vectors written
    independently of the oracle after torchvision's CPU kernel and its test-suite's expected_fn
    (tests/golden/make_deform_golden.py).  Also checks that the fixture really visits the boundary branches."""
    g = golden("deform_conv_kat")
    for tag in "ab":
        a = {k: T(g[tag + "_" + k]) for k in ("x", "offset", "mask", "weight", "bias", "out")}
        out = O.deform_conv2d(a["x"], a["offset"], a["mask"], a["weight"], a["bias"])
        close(out, g[tag + "_out"], 5e-6, rtol=1e-5, what="deform_conv2d vs independent float64 vectors (%s)" % tag)
        N, _, H, W = a["x"].shape
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        hs = torch.stack([ys - 1 + t // 3 + a["offset"][:, 2 * t] for t in range(9)], 1)
        ws = torch.stack([xs - 1 + t % 3 + a["offset"][:, 2 * t + 1] for t in range(9)], 1)
        for pos, size in ((hs, H), (ws, W)):
            assert int(((pos > -1) & (pos < 0)).sum()) > 20 and int(((pos > size - 1) & (pos < size)).sum()) > 20
            assert int((pos == -1).sum()) > 3 and int((pos == size).sum()) > 3 and int((pos == size - 1).sum()) > 3
            assert int((pos == pos.floor()).sum()) > 20 and int((pos > size + 1).sum()) > 3


def test_deform_conv_backward_vs_independent_known_answers(golden):
    """Autograd THROUGH the oracle's deform_conv2d (float64) against tests/golden/deform_conv_backward_kat.npz: the five
    gradients written as float64 scalar loops after torchvision's CPU backward kernels (deformable_col2im,
    deformable_col2im_coord + get_coordinate_weight; tests/golden/make_deform_backward_golden.py) - independent of the
    oracle, which the GPU backward was so far compared with alone.  Checks the fixture's boundary coverage too."""
    g = golden("deform_conv_backward_kat")
    for tag in "abc":
        a = {k: T(g[tag + "_" + k]).double() for k in ("x", "offset", "mask", "weight", "dy")}
        leaves = [a[k].clone().requires_grad_() for k in ("x", "offset", "mask", "weight")]
        bias = torch.zeros(a["weight"].shape[0], dtype=torch.float64, requires_grad=True)
        O.deform_conv2d(leaves[0], leaves[1], leaves[2], leaves[3], bias).backward(a["dy"])
        for leaf, name in zip(leaves + [bias], ("dx", "doffset", "dmask", "dweight", "dbias")):
            want = T(g[tag + "_" + name]).double()
            rms = float(want.pow(2).mean().sqrt())
            err = float((leaf.grad - want).abs().max())
            assert err <= 2e-6 * rms + 1e-7, ("oracle autograd vs independent float64 vectors", tag, name, err, rms)
        N, _, H, W = a["x"].shape
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
        hs = torch.stack([ys - 1 + t // 3 + a["offset"][:, 2 * t] for t in range(9)], 1)
        ws = torch.stack([xs - 1 + t % 3 + a["offset"][:, 2 * t + 1] for t in range(9)], 1)
        for pos, size in ((hs, H), (ws, W)):
            assert int(((pos > -1) & (pos < 0)).sum()) > 10 and int(((pos > size - 1) & (pos < size)).sum()) > 10
            assert int((pos == pos.floor()).sum()) > 10 and int((pos > size + 1).sum()) > 3 and int((pos < -1).sum()) > 3
            assert int((pos == -1).sum()) == 0 and int((pos == size).sum()) == 0      # (measure-zero positions: left out)
