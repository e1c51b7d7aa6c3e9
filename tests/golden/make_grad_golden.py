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
Fixtures hold tensors only; the GPU box never needs /root/reference."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
from make_golden import import_reference, make_sequence, make_state_dict, normalize, npy  # noqa: E402

H, W, N_FRAMES, SEED = 128, 256, 4, 1000
MAX_SAMPLE = 4096


def flow_gts(n, H, W, seed=4321):
    g = torch.Generator().manual_seed(seed)
    return [3.0 * torch.randn(1, 2, H, W, generator=g) for _ in range(n)]


def main():
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
    frames = [normalize(f) for f in make_sequence(SEED, N_FRAMES, H, W)]
    gts = flow_gts(N_FRAMES - 2, H, W)
    outs = model(images=frames, test_mode=False)
    loss, metrics = sequence_loss_acc(outs, gts)
    loss.backward()
    g = {"H": H, "W": W, "seed": SEED, "n_frames": N_FRAMES, "gt_seed": 4321, "loss": np.float64(loss.item()),
         "epe": np.float64(metrics["epe"])}
    for k, o in enumerate(outs):
        g["out%d" % k] = npy(o[:, :, ::4, ::4])
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
    np.savez_compressed(os.path.join(HERE, "accflow_grad_c1.npz"), **g)
    print("loss", loss.item(), "params", len(names), "bytes", os.path.getsize(os.path.join(HERE, "accflow_grad_c1.npz")))
    for n in names[:6] + names[-4:]:
        print(n, g["l2/" + n])


if __name__ == "__main__":
    main()
