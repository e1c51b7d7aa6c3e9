"""Gradient fixture of one training step, made by running the REFERENCE (mulns/AccFlow at /root/reference, read-only) with
its own autograd on the CPU in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_grad_golden.py

What it does (imports and the torchvision stub as in make_golden.py - the stub's forward is the oracle's differentiable
restatement of deform_conv2d, so autograd differentiates exactly the formula the kernels' backward implements):
  * builds the reference AccFlow(RAFT) with the build's deterministic weights, model.train() as train_acc.py:169 does but
    with the frozen estimator left in eval() (its BatchNorm keeps the running statistics: see accflow_amd/train.py), fp32
    (mixed_precision = False: the fixture pins the graph, autocast is a lower-precision evaluation of it);
  * 4 synthetic frames of 128 x 256 -> two fusion steps (both branches of AccFlow.iter, AccFlow_.py:183-191), seeded
    ground-truth flows, loss = loss.sequence_loss_acc (loss.py:30-36), loss.backward();
  * stores the loss, the two predictions (sub-sampled) and for every trainable parameter: L2 norm and sum of its gradient
    (float64) and a strided sample of <= 4096 gradient values.
Cases (VERDICT r04 #4a: pin the training slice where it is benchmarked, not only at C1 size):
    (default)  accflow_grad_c1.npz     4 frames, 128 x 256, batch 1: two fusion steps
    --train    accflow_grad_train.npz  7 frames, 256 x 256, batch 2 (configs/AccRAFT-CVO.yml's crop): five fusion steps,
                                       batch > 1 - the shape family of the benchmarked step (step-mode backward, slice-filling
                                       operators, the LDS form of the deformable backward: 32 x 32 = 1024-pixel planes)
    --big      accflow_grad_big.npz    3 frames, 768 x 768, batch 1: 96 x 96 = 9216-pixel coarse planes, beyond the 4096 the LDS
                                       form of the deformable convolution's backward holds - its atomic fallback
Fixtures hold tensors only; the GPU box never needs /root/reference."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
from make_golden import import_reference, make_sequence, make_state_dict, normalize, npy  # noqa: E402

MAX_SAMPLE = 4096
CASES = {"c1": (128, 256, 4, 1, "accflow_grad_c1.npz"), "train": (256, 256, 7, 2, "accflow_grad_train.npz"),
         "big": (768, 768, 3, 1, "accflow_grad_big.npz")}
SEED = 1000


def flow_gts(n, H, W, seed=4321, batch=1):
    g = torch.Generator().manual_seed(seed)
    return [3.0 * torch.randn(batch, 2, H, W, generator=g) for _ in range(n)]


def main():
    case = "train" if "--train" in sys.argv else "big" if "--big" in sys.argv else "c1"
    H, W, N_FRAMES, BATCH, fname = CASES[case]
    torch.set_num_threads(8)
    build, AccFlow = import_reference()
    from loss import sequence_loss_acc
    ofe = build("acc|raft")
    for p in ofe.parameters():
        p.requires_grad = False                      # train_acc.py:163-164
    model = AccFlow(ofe)
    model.load_state_dict(make_state_dict(model), strict=True)
    model.train()
    model.ofe.eval()
    model.mixed_precision = False
    seqs = [[normalize(f) for f in make_sequence(SEED + b, N_FRAMES, H, W)] for b in range(BATCH)]
    frames = [torch.cat([s[t] for s in seqs], dim=0) for t in range(N_FRAMES)]     # batch item b = sequence SEED + b
    gts = flow_gts(N_FRAMES - 2, H, W, batch=BATCH)
    # (train / big) record the frozen estimator's 1/8-resolution flows as AccFlow.iter consumes them (AccFlow_.py:183-190):
    # the test injects them, so that the heads - the part that carries gradients - see bit-identical inputs on both sides.
    # Without that, the build's estimator (EPE ~1e-5 px from the reference at these sizes) moves enough pre-activations
    # across a ReLU kink to put isolated 1 %-errors on single gradient elements (measured: relative L2 2e-4 instead of 5e-6).
    import networks.AccFlow_ as ref_mod
    recorded = []
    orig_down = ref_mod.downflow8

    def down_rec(flow, *a, **k):
        r = orig_down(flow, *a, **k)
        recorded.append(r.detach().clone())
        return r
    ref_mod.downflow8 = down_rec
    outs = model(images=frames, test_mode=False)
    ref_mod.downflow8 = orig_down
    loss, metrics = sequence_loss_acc(outs, gts)
    loss.backward()
    g = {"H": H, "W": W, "seed": SEED, "n_frames": N_FRAMES, "batch": BATCH, "gt_seed": 4321, "loss": np.float64(loss.item()),
         "epe": np.float64(metrics["epe"])}
    for k, o in enumerate(outs):
        g["out%d" % k] = npy(o[:, :, ::(4 if case == "c1" else 8), ::(4 if case == "c1" else 8)])
    if case != "c1":
        for k, r in enumerate(recorded):          # call k belongs to fusion step i = k + 2
            i = k + 2
            parts = r.chunk(3) if k == 0 else r.chunk(2)
            keys = [(i, i - 1), (i, 0)] + ([(i - 1, 0)] if k == 0 else [])
            for (a_, b_), t_ in zip(keys, parts):
                g["small/%d_%d" % (a_, b_)] = npy(t_)
    names = []
    for name, p in model.named_parameters():
        if name.startswith("ofe."):
            assert p.grad is None
            continue
        assert p.grad is not None, name
        gr = p.grad.detach().double().reshape(-1)
        step = max(1, gr.numel() // MAX_SAMPLE)
        names.append(name)
        g["l2/" + name] = np.float64(gr.norm().item())
        g["sum/" + name] = np.float64(gr.sum().item())
        g["step/" + name] = np.int64(step)
        g["val/" + name] = gr[::step].float().numpy()
    g["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, fname), **g)
    print(case, "loss", loss.item(), "params", len(names), "bytes", os.path.getsize(os.path.join(HERE, fname)))
    for n in names[:6] + names[-4:]:
        print(n, g["l2/" + n])


if __name__ == "__main__":
    main()
