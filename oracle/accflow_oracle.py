"""CPU oracle for the AccFlow inference hot path -- TEST INFRASTRUCTURE ONLY.

A functional, state_dict-driven fp32 restatement (PyTorch CPU tensors, explicit index arithmetic for
every gather-type op) of the reference algorithm, each function citing the reference file:line it
follows (paths relative to the mulns/AccFlow root).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; nothing under accflow_amd/ does, and the product
path has no CPU fallback.

Pinning: the reference has no tests or golden vectors.  This oracle is pinned
against outputs of the reference itself, imported in the build container and run on CPU (fp32, since
torch.cuda.amp.autocast self-disables without CUDA): tests/golden/make_golden.py generated
tests/golden/*.npz and tests/test_oracle_golden.py checks every function here against them.
EXCEPTION - `deform_conv2d`: the reference delegates to torchvision.ops.DeformConv2d (AccFlow_.py:4,83,104;
torchvision 0.16.1 pinned in environment.yml:160), which is absent from this image and not vendored,
so that one function restates torchvision's published modulated deform_conv2d algorithm.  It is pinned by an
INDEPENDENT second restatement: tests/golden/make_deform_golden.py is a separately written float64 scalar-loop
transcription of torchvision's CPU kernel (deformable_im2col + bilinear_interpolate) that shares no code with this
module; its vectors (tests/golden/deform_conv_kat.npz, offsets steered onto every boundary branch) are checked by
tests/test_oracle_golden.py::test_deform_conv_vs_independent_known_answers, beside the identity tests (zero offsets ==
conv2d, integer offsets == shifted conv, mask 0 == bias, half-pixel offsets == grid_sample).  What has never executed
offline is torchvision's own binary.
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# small helpers


def _conv(x, sd, name, stride=1, padding=0):
    return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=stride, padding=padding)


def _bn_eval(x, sd, name, eps=1e-5):
    """nn.BatchNorm2d in eval mode (running statistics)."""
    w, b = sd[name + ".weight"], sd[name + ".bias"]
    m, v = sd[name + ".running_mean"], sd[name + ".running_var"]
    return (x - m[None, :, None, None]) / torch.sqrt(v[None, :, None, None] + eps) * w[None, :, None, None] \
        + b[None, :, None, None]


def _instance_norm(x, eps=1e-5):
    """nn.InstanceNorm2d defaults: per (n, c) plane, biased variance, no affine (extractor.py:36-39)."""
    m = x.mean(dim=(2, 3), keepdim=True)
    v = ((x - m) ** 2).mean(dim=(2, 3), keepdim=True)
    return (x - m) / torch.sqrt(v + eps)


def _norm(x, sd, name, kind):
    if kind == "instance":
        return _instance_norm(x)
    if kind == "batch":
        return _bn_eval(x, sd, name)
    if kind == "none":
        return x
    raise ValueError(kind)


# ----------------------------------------------------------------------------------------------
# R1  BasicEncoder / ResidualBlock  (raft/extractor.py:5-63, :115-225)


def residual_block(x, sd, p, kind, stride):
    y = torch.relu(_norm(_conv(x, sd, p + ".conv1", stride=stride, padding=1), sd, p + ".norm1", kind))
    y = torch.relu(_norm(_conv(y, sd, p + ".conv2", padding=1), sd, p + ".norm2", kind))
    if (p + ".downsample.0.weight") in sd:
        x = _norm(_conv(x, sd, p + ".downsample.0", stride=stride), sd, p + ".norm3", kind)
    return torch.relu(x + y)


def basic_encoder(x, sd, p, kind):
    """x: (N,3,H,W) -> (N,out,H/8,W/8)   (extractor.py:201-225)"""
    x = torch.relu(_norm(_conv(x, sd, p + ".conv1", stride=2, padding=3), sd, p + ".norm1", kind))
    for li, st in ((1, 1), (2, 2), (3, 2)):
        x = residual_block(x, sd, "%s.layer%d.0" % (p, li), kind, st)
        x = residual_block(x, sd, "%s.layer%d.1" % (p, li), kind, 1)
    return _conv(x, sd, p + ".conv2")


# ----------------------------------------------------------------------------------------------
# R2-R4  correlation volume, pyramid, lookup  (raft/corr.py)


def corr_volume(fmap1, fmap2):
    """(B,C,H,W)x2 -> (B,H,W,1,H,W):  <f1[:,i], f2[:,j]> / sqrt(C)   (corr.py:47-55)"""
    B, C, H, W = fmap1.shape
    a = fmap1.reshape(B, C, H * W)
    b = fmap2.reshape(B, C, H * W)
    corr = torch.matmul(a.transpose(1, 2), b)
    return corr.reshape(B, H, W, 1, H, W) / math.sqrt(C)


def corr_pyramid(fmap1, fmap2, num_levels=4):
    """list of (B*H*W, 1, Hl, Wl); each level = 2x2 mean of the previous, floor on odd sizes (corr.py:17-22)"""
    corr = corr_volume(fmap1, fmap2)
    B, H, W = corr.shape[:3]
    lvl = corr.reshape(B * H * W, 1, H, W)
    pyr = [lvl]
    for _ in range(num_levels - 1):
        h2, w2 = lvl.shape[2] // 2, lvl.shape[3] // 2
        t = lvl[:, :, :2 * h2, :2 * w2]
        lvl = (((t[:, :, 0::2, 0::2] + t[:, :, 0::2, 1::2]) + t[:, :, 1::2, 0::2]) + t[:, :, 1::2, 1::2]) * 0.25
        pyr.append(lvl)
    return pyr


def _gather_zeros(plane, yy, xx):
    """plane: (M, H, W); yy, xx: (M, ...) integer indices -> values, 0 outside the plane."""
    M, H, W = plane.shape
    ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
    idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).reshape(M, -1)
    v = torch.gather(plane.reshape(M, H * W), 1, idx).reshape(yy.shape)
    return torch.where(ok, v, torch.zeros_like(v))


def bilinear_zeros(plane, sx, sy):
    """F.grid_sample(bilinear, zeros, align_corners=True) at PIXEL coordinates: per-corner zero padding
    (what bilinear_sampler, raft/utils/utils.py:66-80, evaluates after its normalise round trip)."""
    x0 = torch.floor(sx)
    y0 = torch.floor(sy)
    ax, ay = sx - x0, sy - y0
    x0, y0 = x0.long(), y0.long()
    v00 = _gather_zeros(plane, y0, x0)
    v01 = _gather_zeros(plane, y0, x0 + 1)
    v10 = _gather_zeros(plane, y0 + 1, x0)
    v11 = _gather_zeros(plane, y0 + 1, x0 + 1)
    return v00 * ((1 - ax) * (1 - ay)) + v01 * (ax * (1 - ay)) + v10 * ((1 - ax) * ay) + v11 * (ax * ay)


def corr_lookup(pyramid, coords, radius=4):
    """coords (B,2,H,W) [x,y] -> (B, L*(2r+1)^2, H, W); channel l*81 + i*9 + j samples level l at
    (cx/2^l + (i-r), cy/2^l + (j-r)): i walks x, j walks y (corr.py:30-45: meshgrid(dy,dx) is added to an
    (x,y) centroid)."""
    B, _, H, W = coords.shape
    r = radius
    n = 2 * r + 1
    c = coords.permute(0, 2, 3, 1).reshape(B * H * W, 2)
    d = torch.arange(-r, r + 1, dtype=coords.dtype)
    out = []
    for l, lvl in enumerate(pyramid):
        cx = (c[:, 0] / 2 ** l)[:, None, None] + d[None, :, None]   # (M, i, 1)
        cy = (c[:, 1] / 2 ** l)[:, None, None] + d[None, None, :]   # (M, 1, j)
        sx = cx.expand(-1, n, n)
        sy = cy.expand(-1, n, n)
        v = bilinear_zeros(lvl[:, 0], sx, sy)                        # (M, i, j)
        out.append(v.reshape(B, H, W, n * n))
    return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous()


# ----------------------------------------------------------------------------------------------
# R5-R8  update block  (raft/update.py)


def motion_encoder(flow, corr, sd, p):
    """BasicMotionEncoder.forward (update.py:89-97)"""
    cor = torch.relu(_conv(corr, sd, p + ".convc1"))
    cor = torch.relu(_conv(cor, sd, p + ".convc2", padding=1))
    flo = torch.relu(_conv(flow, sd, p + ".convf1", padding=3))
    flo = torch.relu(_conv(flo, sd, p + ".convf2", padding=1))
    out = torch.relu(_conv(torch.cat([cor, flo], 1), sd, p + ".conv", padding=1))
    return torch.cat([out, flow], 1)


def sep_conv_gru(h, x, sd, p):
    """SepConvGRU.forward (update.py:45-60)"""
    for s, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([h, x], 1)
        z = torch.sigmoid(_conv(hx, sd, p + ".convz" + s, padding=pad))
        r = torch.sigmoid(_conv(hx, sd, p + ".convr" + s, padding=pad))
        q = torch.tanh(_conv(torch.cat([r * h, x], 1), sd, p + ".convq" + s, padding=pad))
        h = (1 - z) * h + z * q
    return h


def flow_head(net, sd, p):
    return _conv(torch.relu(_conv(net, sd, p + ".conv1", padding=1)), sd, p + ".conv2", padding=1)


def mask_head(net, sd, p, scale):
    return scale * _conv(torch.relu(_conv(net, sd, p + ".0", padding=1)), sd, p + ".2")


def update_block(net, inp, corr, flow, sd, p="update_block", attention=None):
    """BasicUpdateBlock.forward (update.py:127-136) / GMAUpdateBlock.forward (gma/update.py:127-139)
    -> (net, mask, delta_flow)"""
    motion = motion_encoder(flow, corr, sd, p + ".encoder")
    if attention is None:
        x = torch.cat([inp, motion], 1)
    else:
        x = torch.cat([inp, motion, gma_aggregate(attention, motion, sd, p + ".aggregator")], 1)
    net = sep_conv_gru(net, x, sd, p + ".gru")
    return net, mask_head(net, sd, p + ".mask", 0.25), flow_head(net, sd, p + ".flow_head")


# ----------------------------------------------------------------------------------------------
# R9  convex upsampling  (raft/raft.py:81-92)


def convex_upsample(flow, mask):
    """out[n,c,8h+a,8w+b] = sum_k softmax_k(mask[n,k*64+a*8+b,h,w]) * 8*flow_zp[n,c,h+k//3-1,w+k%3-1]"""
    N, _, H, W = flow.shape
    m = torch.softmax(mask.reshape(N, 9, 8, 8, H, W), dim=1)
    fp = F.pad(8 * flow, (1, 1, 1, 1))
    out = torch.zeros(N, 2, 8, 8, H, W, dtype=flow.dtype)
    for k in range(9):
        dy, dx = k // 3, k % 3
        nb = fp[:, :, dy:dy + H, dx:dx + W]                      # (N,2,H,W)
        out = out + m[:, k][:, None] * nb[:, :, None, None]
    return out.permute(0, 1, 4, 2, 5, 3).reshape(N, 2, 8 * H, 8 * W)


# ----------------------------------------------------------------------------------------------
# R10  RAFT.forward (raft/raft.py:94-146) and R11/R12 GMA (gma/gma.py:70-125, gma/modules.py)


def coords_grid(B, H, W):
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    return torch.stack([xs, ys], 0).float()[None].repeat(B, 1, 1, 1)


def gma_attention(inp, sd, p="att"):
    """Attention.forward, heads=1, content only (gma/modules.py:54-76) -> (B,1,P,P)"""
    B, C, H, W = inp.shape
    qk = F.conv2d(inp, sd[p + ".to_qk.weight"])
    q, k = qk.chunk(2, dim=1)
    D = q.shape[1]
    q = (D ** -0.5) * q.reshape(B, D, H * W)
    k = k.reshape(B, D, H * W)
    sim = torch.matmul(q.transpose(1, 2), k)
    return torch.softmax(sim, dim=-1)[:, None]


def gma_aggregate(attn, fmap, sd, p):
    """Aggregate.forward (gma/modules.py:102-115): fmap + gamma * (attn @ v)"""
    B, C, H, W = fmap.shape
    v = F.conv2d(fmap, sd[p + ".to_v.weight"]).reshape(B, C, H * W)          # (B, d, j)
    out = torch.matmul(attn[:, 0], v.transpose(1, 2))                          # (B, i, d)
    out = out.transpose(1, 2).reshape(B, C, H, W)
    return fmap + sd[p + ".gamma"] * out


def raft_forward(sd, image1, image2, iters=12, flow_init=None, gma=False, return_all=False, trace=None):
    """-> flow_up (N,2,H,W).  trace: optional dict filled with named intermediates."""
    fmaps = basic_encoder(torch.cat([image1, image2], 0), sd, "fnet", "instance")
    B = image1.shape[0]
    fmap1, fmap2 = fmaps[:B], fmaps[B:]
    pyr = corr_pyramid(fmap1, fmap2)
    cnet = basic_encoder(image1, sd, "cnet", "batch")
    net, inp = torch.tanh(cnet[:, :128]), torch.relu(cnet[:, 128:])
    attention = gma_attention(inp, sd) if gma else None
    H8, W8 = fmap1.shape[2:]
    coords0 = coords_grid(B, H8, W8)
    coords1 = coords0.clone()
    if flow_init is not None:
        coords1 = coords1 + flow_init
    if trace is not None:
        trace.update(fmap1=fmap1, fmap2=fmap2, cnet=cnet, pyramid=pyr, attention=attention)
    ups = []
    for itr in range(iters):
        corr = corr_lookup(pyr, coords1)
        flow = coords1 - coords0
        net, up_mask, delta = update_block(net, inp, corr, flow, sd, attention=attention)
        coords1 = coords1 + delta
        if trace is not None and itr == 0:
            trace.update(corr0=corr, net1=net, mask1=up_mask, delta1=delta)
        if return_all or itr == iters - 1:
            ups.append(convex_upsample(coords1 - coords0, up_mask))
    if trace is not None:
        trace.update(flow_small=coords1 - coords0)
    return ups if return_all else ups[-1]


# ----------------------------------------------------------------------------------------------
# R14-R19  AccFlow pieces  (networks/AccFlow_.py, networks/utils.py, networks/modules.py)


def backwarp(img, flow):
    """out[n,c,y,x] = bilinear_zeros(img[n,c], x+u, y+v)   (networks/utils.py:96-124)"""
    N, C, H, W = img.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=img.dtype), torch.arange(W, dtype=img.dtype), indexing="ij")
    sx = (xs[None] + flow[:, 0])[:, None].expand(N, C, H, W).reshape(N * C, H, W)
    sy = (ys[None] + flow[:, 1])[:, None].expand(N, C, H, W).reshape(N * C, H, W)
    return bilinear_zeros(img.reshape(N * C, H, W), sx, sy).reshape(N, C, H, W)


def get_occ(F12, I1, I2, binary=True):
    """AccFlow_.py:127-135"""
    e = torch.abs(I1 - backwarp(I2, F12))
    if binary:
        e = e.mean(dim=1, keepdim=True)
        return torch.where(e <= 1.0, torch.ones_like(e), torch.zeros_like(e))
    return e


def get_occ_error(F12, I1, I2):
    """the mean abs error map getOcc thresholds at 1.0 (used by tests to excuse threshold ties)"""
    return torch.abs(I1 - backwarp(I2, F12)).mean(dim=1, keepdim=True)


def downflow8(flow):
    """F.interpolate(size=(H/8,W/8), bilinear, align_corners=True) / 8   (AccFlow_.py:138-142)"""
    N, C, H, W = flow.shape
    assert H % 8 == 0 and W % 8 == 0
    h, w = H // 8, W // 8
    sy = torch.arange(h, dtype=torch.float32) * (float(H - 1) / float(h - 1) if h > 1 else 0.0)
    sx = torch.arange(w, dtype=torch.float32) * (float(W - 1) / float(w - 1) if w > 1 else 0.0)
    y0, x0 = sy.long(), sx.long()
    y1 = torch.where(y0 < H - 1, y0 + 1, y0)
    x1 = torch.where(x0 < W - 1, x0 + 1, x0)
    ly, lx = (sy - y0)[:, None], (sx - x0)[None, :]
    f = flow
    top = (1 - lx) * f[:, :, y0][:, :, :, x0] + lx * f[:, :, y0][:, :, :, x1]
    bot = (1 - lx) * f[:, :, y1][:, :, :, x0] + lx * f[:, :, y1][:, :, :, x1]
    return ((1 - ly) * top + ly * bot) / 8


def flow_encoder(x, sd, p="flow_encoder"):
    """AccFlow_.py:56-65"""
    x = torch.relu(_conv(x, sd, p + ".conv1", padding=3))
    x = torch.relu(_conv(x, sd, p + ".conv2", padding=1))
    return _conv(x, sd, p + ".conv3")


def deform_conv2d(x, offset, mask, weight, bias):
    """torchvision.ops.deform_conv2d, modulated (v2), 3x3 / stride 1 / pad 1 / dilation 1 / 1 group / 1
    offset group -- pinned by the independent float64 vectors of tests/golden/deform_conv_kat.npz (see the module
    docstring; torchvision's own binary is not in the image).  For output (y,x) and tap t = ky*3+kx the input
    is sampled at (y-1+ky + offset[2t], x-1+kx + offset[2t+1]); a sample is 0 when h<=-1, h>=H, w<=-1 or
    w>=W, else bilinear with zero corners outside; times mask[t]; columns ordered c*9+t against
    weight.view(Cout, Cin*9)."""
    N, C, H, W = x.shape
    Cout = weight.shape[0]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=x.dtype), torch.arange(W, dtype=x.dtype), indexing="ij")
    cols = []
    xf = x.reshape(N * C, H, W)
    for t in range(9):
        ky, kx = t // 3, t % 3
        h = (ys[None] - 1 + ky + offset[:, 2 * t])            # (N,H,W)
        w = (xs[None] - 1 + kx + offset[:, 2 * t + 1])
        inside = (h > -1) & (h < H) & (w > -1) & (w < W)
        hh = h[:, None].expand(N, C, H, W).reshape(N * C, H, W)
        ww = w[:, None].expand(N, C, H, W).reshape(N * C, H, W)
        v = bilinear_zeros(xf, ww, hh).reshape(N, C, H, W)
        v = torch.where(inside[:, None], v, torch.zeros_like(v))
        cols.append(v * mask[:, t][:, None])
    col = torch.stack(cols, dim=2).reshape(N, C * 9, H * W)     # index c*9 + t
    out = torch.matmul(weight.reshape(Cout, C * 9)[None], col).reshape(N, Cout, H, W)
    return out + bias[None, :, None, None]


def accplus(df, f, o, c, sd, p="accplus"):
    """AccPlus.forward (AccFlow_.py:97-109); ZeroConv2d: conv * exp(3*scale) (modules.py:94-96)"""
    x = torch.cat([df, f, o], 1)
    x = _conv(torch.relu(_conv(x, sd, p + ".conv1.0", padding=1)), sd, p + ".conv1.2", padding=1)
    x = torch.cat([x, c], 1)
    x = torch.relu(_conv(x, sd, p + ".conv2.0", padding=1))
    x = torch.relu(_conv(x, sd, p + ".conv2.2", padding=1))
    x = _conv(x, sd, p + ".conv2.4.conv", padding=1) * torch.exp(sd[p + ".conv2.4.scale"] * 3)
    off, m = x[:, :18], torch.sigmoid(x[:, 18:27])
    f_ = deform_conv2d(f, off, m, sd[p + ".dconv.weight"], sd[p + ".dconv.bias"])
    x = torch.cat([f_, df, o], 1)
    x = _conv(torch.relu(_conv(x, sd, p + ".conv3.0", padding=1)), sd, p + ".conv3.2", padding=1)
    x = torch.cat([x, c, f_, df], 1)
    x = torch.relu(_conv(x, sd, p + ".conv4.0", padding=1))
    x = torch.relu(_conv(x, sd, p + ".conv4.2", padding=1))
    return _conv(x, sd, p + ".conv4.4"), dict(off=off, m=m, f_=f_)


def blending(f1, f2, emap, sd, p="blending"):
    """AccFlow_.py:122-124"""
    m = torch.sigmoid(_conv(torch.relu(_conv(emap, sd, p + ".mask.0")), sd, p + ".mask.2", padding=1))
    return f1 * m + (1 - m) * f2


def flow_decoder(x, sd, p="flow_decoder"):
    """AccFlow_.py:40-45 (no 0.25 on the mask)"""
    flow_small = _conv(torch.relu(_conv(x, sd, p + ".flow.0", padding=1)), sd, p + ".flow.2", padding=1)
    mask = mask_head(x, sd, p + ".mask", 1.0)
    return flow_small, convex_upsample(flow_small, mask)


def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def accflow_fuse(sd, dflow, flow_ini, F2n, c1, c2, cn, trace=None):
    """AccFlow.iter after the estimator calls (AccFlow_.py:191-201)"""
    enc = flow_encoder(torch.cat([flow_ini, dflow, F2n], 0), sd)
    N = dflow.shape[0]
    f_ini, df, f = enc[:N], enc[N:2 * N], enc[2 * N:]
    o = get_occ(dflow, c1, c2)
    f_acc, extra = accplus(df, f, o, c1, sd)
    emap = get_occ(flow_ini, c1, cn, binary=False)
    f_fuse = blending(f_ini, f_acc, emap, sd)
    out_small, out = flow_decoder(f_fuse, sd)
    if trace is not None:
        trace.update(f_ini=f_ini, df=df, f=f, o=o, f_acc=f_acc, emap=emap, f_fuse=f_fuse, out_small=out_small,
                     o_err=get_occ_error(dflow, c1, c2), **extra)
    return out_small, out


def accflow_forward(sd, images, iters=12, gma=False, trace=None):
    """AccFlow.forward (AccFlow_.py:157-175): [F(2->0), ..., F(n-1->0)]; estimator weights under 'ofe.'"""
    ofe = _sub(sd, "ofe.")
    outs, F2n = [], None
    In = images[0]
    for i in range(2, len(images)):
        I1, I2 = images[i], images[i - 1]
        if F2n is None:
            flows = downflow8(raft_forward(ofe, torch.cat([I1, I1, I2]), torch.cat([I2, In, In]), iters, gma=gma))
            dflow, flow_ini, F2n = flows.chunk(3)
        else:
            flows = downflow8(raft_forward(ofe, torch.cat([I1, I1]), torch.cat([I2, In]), iters, gma=gma))
            dflow, flow_ini = flows.chunk(2)
        ctx = basic_encoder(torch.cat([I1, I2, In], 0), sd, "context", "none")
        N = I1.shape[0]
        tr = {} if trace is not None else None
        F2n, up = accflow_fuse(sd, dflow, flow_ini, F2n, ctx[:N], ctx[N:2 * N], ctx[2 * N:], trace=tr)
        if trace is not None:
            tr.update(dflow=dflow, flow_ini=flow_ini, c1=ctx[:N], c2=ctx[N:2 * N], cn=ctx[2 * N:])
            trace["step%d" % i] = tr
        outs.append(up)
    return outs


def compose_flow(step, acc):
    """flow a -> c from step = a -> b and acc = b -> c (same resolution): step + backwarp(acc, step)"""
    return step + backwarp(acc, step)


def accflow_forward_warm(sd, images, iters=12, warm_iters=None, gma=False):
    """Warm-start schedule of the build's AccFlow(warm_start=True) (the reference only provides the
    mechanism, RAFT's `flow_init`, raft.py:123-124, and lists "Add warmstart mode" as a TODO in README.md:11): the
    adjacent pairs i -> i-1 are estimated cold; the long-range estimate i -> 0 of step i starts from
    flow_init = F(i->i-1) (+) F_acc(i-1->0) at 1/8 resolution and runs warm_iters iterations; fusion as
    accflow_forward."""
    ofe = _sub(sd, "ofe.")
    n = len(images)
    warm_iters = iters if warm_iters is None else warm_iters
    adj = {i: downflow8(raft_forward(ofe, images[i], images[i - 1], iters, gma=gma)) for i in range(1, n)}
    N = images[0].shape[0]
    outs, F2n = [], adj[1]
    for i in range(2, n):
        seed = compose_flow(adj[i], F2n)
        flow_ini = downflow8(raft_forward(ofe, images[i], images[0], warm_iters, flow_init=seed, gma=gma))
        ctx = basic_encoder(torch.cat([images[i], images[i - 1], images[0]], 0), sd, "context", "none")
        F2n, up = accflow_fuse(sd, adj[i], flow_ini, F2n, ctx[:N], ctx[N:2 * N], ctx[2 * N:])
        outs.append(up)
    return outs


# ----------------------------------------------------------------------------------------------
# R20  evaluation harness pieces  (test_cvo.py:32-101)


def preprocess_images(imgs_0_255):
    """test_cvo.py:41-42: 2*(x/255)-1, split into 3-channel frames"""
    return list((2 * (imgs_0_255 / 255.0) - 1).split(3, dim=1))


def calc_occ_mask(bflow, fflow):
    """test_cvo.py:53-78 (note `length_sq` returns the L2 norm, :63-66) -> (occ_bw, occ_fw)"""
    def norm(x):
        return torch.pow(torch.sum(x ** 2, dim=1, keepdim=True), 0.5)
    mag = norm(fflow) + norm(bflow)
    diff_fw = fflow + backwarp(bflow, fflow)
    diff_bw = bflow + backwarp(fflow, bflow)
    thr = 0.01 * mag + 0.5
    return (norm(diff_bw) > thr).float(), (norm(diff_fw) > thr).float()


def cal_epe(pred, label, occ_mask):
    """test_cvo.py:81-101 -> (epe_all, epe_occ, epe_vis), each (N,)"""
    diff = torch.norm(pred - label, p=2, dim=1, keepdim=True)
    epe_all = diff.mean(dim=(1, 2, 3))
    epe_occ = (diff * occ_mask).sum(dim=(1, 2, 3)) / occ_mask.sum(dim=(1, 2, 3))
    epe_vis = (diff * (1 - occ_mask)).sum(dim=(1, 2, 3)) / (1 - occ_mask).sum(dim=(1, 2, 3))
    return epe_all, epe_occ, epe_vis


def epe(a, b):
    """mean / max end-point error between two flow fields (formula of test_cvo.py:92-93)"""
    d = torch.norm(a - b, p=2, dim=1)
    return float(d.mean()), float(d.max())
