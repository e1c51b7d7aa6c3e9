"""Generate the golden fixtures by running the REFERENCE implementation (mulns/AccFlow, mounted
read-only at /root/reference) on CPU in the build container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--big]

What it does
  * imports the reference's networks package (torchvision is absent from the image, so a stub
    `torchvision.ops.DeformConv2d` with torchvision's parameter names is injected whose forward calls the
    oracle's restatement of deform_conv2d - the one op whose reference arithmetic cannot be executed here);
  * builds the reference RAFT / RAFTGMA / AccFlow(RAFT) and loads the build's deterministic, name-keyed
    weights with strict=True (which also proves state_dict compatibility);
  * runs them on the build's deterministic synthetic frames and stores inputs-by-seed + outputs (and
    sub-sampled intermediates) as .npz next to this script.
Nothing from the reference is copied: fixtures hold tensors only.  The GPU box never needs /root/reference.
"""
import argparse
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

sys.path.insert(0, ROOT)
from accflow_amd.data.synthetic import make_sequence, make_state_dict, normalize  # noqa: E402
from oracle import accflow_oracle as O  # noqa: E402


def import_reference():
    # stub torchvision.ops.DeformConv2d (same ctor signature / parameter names as torchvision 0.16)
    class DeformConv2d(torch.nn.Module):
        def __init__(self, cin, cout, k, s, p):
            super().__init__()
            self.weight = torch.nn.Parameter(torch.zeros(cout, cin, k, k))
            self.bias = torch.nn.Parameter(torch.zeros(cout))

        def forward(self, x, offset, mask):
            return O.deform_conv2d(x, offset, mask, self.weight, self.bias)

    tv = types.ModuleType("torchvision")
    tvo = types.ModuleType("torchvision.ops")
    tvo.DeformConv2d = DeformConv2d
    tv.ops = tvo
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.ops"] = tvo
    # our repo root also has a `networks` shim: make sure the reference's package wins in THIS process
    for k in [k for k in sys.modules if k == "networks" or k.startswith("networks.")]:
        del sys.modules[k]
    sys.path.insert(0, REF)
    import networks as ref_networks
    assert ref_networks.__file__.startswith(REF), ref_networks.__file__
    from networks.AccFlow_ import AccFlow
    return ref_networks.build_flow_estimator, AccFlow


def npy(t):
    return t.detach().cpu().numpy().astype(np.float32)


def pair(seed, H, W):
    fr = [normalize(f) for f in make_sequence(seed, 2, H, W)]
    return fr[1], fr[0]  # flow from frame 1 to frame 0


@torch.no_grad()
def golden_raft(build, name, gma, H, W, tag):
    torch.manual_seed(0)
    model = build(name).eval()
    sd = make_state_dict(model)
    model.load_state_dict(sd, strict=True)
    i1, i2 = pair(1000, H, W)
    g = {"H": H, "W": W, "seed": 1000}
    fmap1, fmap2 = model.fnet([i1, i2])
    cnet = model.cnet(i1)
    g["fmap1"], g["fmap2"], g["cnet"] = npy(fmap1), npy(fmap2), npy(cnet)
    from networks.raft.corr import CorrBlock
    cb = CorrBlock(fmap1.float(), fmap2.float(), radius=4)
    P = fmap1.shape[2] * fmap1.shape[3]
    sel = np.arange(0, P, max(1, P // 64))[:64]
    g["pyr_sel"] = sel.astype(np.int64)
    for l in range(4):
        g["pyr%d" % l] = npy(cb.corr_pyramid[l][sel])
    from networks.raft.utils.utils import coords_grid
    B, _, h, w = fmap1.shape
    coords0 = coords_grid(B, h, w, device="cpu")
    gen = torch.Generator().manual_seed(7)
    coords_r = coords0 + 6.0 * torch.randn(coords0.shape, generator=gen)
    g["coords_r"] = npy(coords_r)
    g["lookup0"] = npy(cb(coords0))
    g["lookup_r"] = npy(cb(coords_r))
    net, inp = torch.split(cnet, [128, 128], dim=1)
    net, inp = torch.tanh(net), torch.relu(inp)
    flow_r = coords_r - coords0
    if gma:
        attn = model.att(inp)
        g["attn_rowsum"] = npy(attn.sum(-1))
        rows = np.arange(0, P, max(1, P // 32))[:32]
        g["attn_rows_sel"] = rows.astype(np.int64)
        g["attn_rows"] = npy(attn[:, 0, rows])
        motion = model.update_block.encoder(flow_r, cb(coords_r))
        g["motion"] = npy(motion)
        g["motion_global"] = npy(model.update_block.aggregator(attn, motion))
        net1, mask1, delta1 = model.update_block(net, inp, cb(coords_r), flow_r, attn)
    else:
        g["motion"] = npy(model.update_block.encoder(flow_r, cb(coords_r)))
        net1, mask1, delta1 = model.update_block(net, inp, cb(coords_r), flow_r)
    g["ub_net"], g["ub_mask_s"], g["ub_delta"] = npy(net1), npy(mask1[:, ::9]), npy(delta1)
    g["upsample"] = npy(model.upsample_flow(flow_r + delta1, mask1))
    for it in (1, 4, 12):
        out = model(i1, i2, iters=it)
        g["flow_it%d" % it] = npy(out if it == 12 else out[:, :, ::2, ::2])
    fi = 0.5 * torch.randn(coords0.shape, generator=gen)
    g["flow_init"] = npy(fi)
    g["flow_it4_init"] = npy(model(i1, i2, iters=4, flow_init=fi)[:, :, ::2, ::2])
    if gma:  # encoders / pyramid / lookup are the same code as RAFT's and already pinned by raft_c1
        for k in ("fmap1", "fmap2", "pyr_sel", "pyr0", "pyr1", "pyr2", "pyr3", "lookup0", "lookup_r", "upsample"):
            g.pop(k)
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **g)
    print(tag, {k: v.shape for k, v in g.items() if hasattr(v, "shape")})
    return model


@torch.no_grad()
def golden_accflow(build, AccFlow, H, W, n_frames, tag, full, ofe="acc|raft"):
    model = AccFlow(build(ofe).eval()).eval()
    sd = make_state_dict(model)
    model.load_state_dict(sd, strict=True)
    frames = [normalize(f) for f in make_sequence(1000, n_frames, H, W)]
    g = {"H": H, "W": W, "seed": 1000, "n_frames": n_frames}
    cap = {}
    if full:
        def hook(name):
            def fn(mod, inp, out):
                if name in cap:
                    return
                cap[name] = out
                if name == "accplus":
                    cap["accplus_in"] = inp
            return fn
        hs = [getattr(model, n).register_forward_hook(hook(n))
              for n in ("flow_encoder", "context", "accplus", "blending", "flow_decoder")]
    outs = model(images=frames, test_mode=False)
    for k, o in enumerate(outs):
        g["out%d" % k] = npy(o if full else o[:, :, ::8, ::8])
    if full:
        for h in hs:
            h.remove()
        f_ini, df, f = cap["flow_encoder"]
        c1, c2, cn = cap["context"]
        g.update(s2_f_ini=npy(f_ini), s2_f=npy(f), s2_c1=npy(c1), s2_cn=npy(cn[:, ::4]))
        g["s2_o"] = npy(cap["accplus_in"][2])
        g["s2_f_acc"] = npy(cap["accplus"])
        g["s2_f_fuse"] = npy(cap["blending"])
        g["s2_out_small"] = npy(cap["flow_decoder"][0])
        # the three 1/8-res estimator flows of step 2, recomputed exactly as AccFlow.iter does (:184-185)
        from networks.AccFlow_ import downflow8
        I1, I2, In = frames[2], frames[1], frames[0]
        fl = downflow8(model.ofe(torch.cat([I1, I1, I2]), torch.cat([I2, In, In])))
        g["s2_dflow"], g["s2_flow_ini"], g["s2_F2n"] = [npy(t) for t in fl.chunk(3)]
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **g)
    print(tag, {k: v.shape for k, v in g.items() if hasattr(v, "shape")})


@torch.no_grad()
def golden_harness():
    """calc_occ_mask / cal_epe: exec test_cvo.py:53-101 (the file itself cannot be imported: module-level
    argparse + .cuda())."""
    src = open(os.path.join(REF, "test_cvo.py")).read().split("\n")
    code = "\n".join(src[52:101])
    from networks.utils import backwarp
    ns = {"torch": torch, "backwarp": backwarp}
    exec(compile(code, "test_cvo_53_101", "exec"), ns)
    gen = torch.Generator().manual_seed(11)
    fflow = torch.randn(2, 2, 32, 48, generator=gen) * 3
    bflow = -fflow + 0.4 * torch.randn(2, 2, 32, 48, generator=gen)
    pred = bflow + 0.7 * torch.randn(2, 2, 32, 48, generator=gen)
    occ_bw, occ_fw = ns["calc_occ_mask"](bflow, fflow)
    e_all, e_occ, e_vis = ns["cal_epe"](pred, bflow, occ_bw)
    img = torch.randn(2, 5, 32, 48, generator=gen)
    from networks.AccFlow_ import downflow8
    big = torch.randn(2, 2, 64, 96, generator=gen) * 4
    g = dict(fflow=npy(fflow), bflow=npy(bflow), pred=npy(pred), occ_bw=npy(occ_bw), occ_fw=npy(occ_fw),
             epe_all=npy(e_all), epe_occ=npy(e_occ), epe_vis=npy(e_vis), img=npy(img),
             warped=npy(backwarp(img, fflow)), big=npy(big), down=npy(downflow8(big)))
    np.savez_compressed(os.path.join(HERE, "harness.npz"), **g)
    print("harness", {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true", help="also 480x1024 fixtures (minutes of CPU)")
    ap.add_argument("--c5", action="store_true", help="only BASELINE configs[4]: AccFlow(GMA) 7x720x1280 (several minutes, ~20 GB)")
    a = ap.parse_args()
    torch.set_num_threads(8)
    build, AccFlow = import_reference()
    if a.c5:
        golden_accflow(build, AccFlow, 720, 1280, 7, "accflow_gma_c5", full=False, ofe="acc|gma")
        sys.exit(0)
    golden_harness()
    golden_raft(build, "raft", False, 128, 256, "raft_c1")
    golden_raft(build, "gma", True, 128, 256, "gma_c1")
    golden_accflow(build, AccFlow, 128, 256, 4, "accflow_c1", full=True)
    if a.big:
        torch.manual_seed(0)
        m = build("raft").eval()
        m.load_state_dict(make_state_dict(m), strict=True)
        i1, i2 = pair(1000, 480, 1024)
        with torch.no_grad():
            out = m(i1, i2, iters=12)
        np.savez_compressed(os.path.join(HERE, "raft_c2.npz"), H=480, W=1024, seed=1000, flow_it12_s8=npy(out[:, :, ::8, ::8]))
        print("raft_c2 done")
        golden_accflow(build, AccFlow, 480, 1024, 7, "accflow_c3", full=False)
