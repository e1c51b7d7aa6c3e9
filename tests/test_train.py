"""One training step (accflow_amd/train.py) against the REFERENCE's own autograd: the fixture
tests/golden/accflow_grad_*.npz hold the loss, the predictions and every trainable parameter's gradient (norm, sum and a
strided sample) of train_acc.py's loss on seeded sequences, produced by tests/golden/make_grad_golden.py from /root/reference
in fp32 on the CPU (the reference run with `ofe.eval()` and `mixed_precision = False`: the frozen estimator's BatchNorm on
its running statistics, the graph in fp32 - accflow_amd/train_acc.py lists both as deliberate differences):
  c1     4 frames, 128 x 256, batch 1 (two fusion steps)
  train  7 frames, 256 x 256, batch 2 (the benchmarked shape family: five steps, batch > 1, the step-mode backward, the
         LDS form of the deformable convolution's backward)
  big    3 frames, 768 x 768, batch 1 (96 x 96 coarse planes: the deformable backward's atomic fallback, > 4096 pixels)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GRAD_TOL = 2e-4      # of the gradient's RMS, per sampled element; both sides accumulate ~10^5-term sums in fp32


def _setup():
    from accflow_amd.data.synthetic import make_sequence, make_state_dict, normalize
    from accflow_amd.networks import build_flow_estimator
    from accflow_amd.networks.AccFlow_ import AccFlow
    model = AccFlow(build_flow_estimator("acc|raft"))
    model.load_state_dict(make_state_dict(model), strict=True)
    return model.cuda().eval(), make_sequence, normalize


def _gts(n, H, W, seed, batch=1):
    g = torch.Generator().manual_seed(seed)
    return [(3.0 * torch.randn(batch, 2, H, W, generator=g)).cuda() for _ in range(n)]


@pytest.mark.parametrize("case", ["c1", "train", "big"])
def test_gradients_match_reference_autograd(golden, case):
    from accflow_amd import train
    G = golden("accflow_grad_" + case)
    H, W, n = int(G["H"]), int(G["W"]), int(G["n_frames"])
    batch = int(G["batch"]) if "batch" in G else 1
    model, make_sequence, normalize = _setup()
    seqs = [[normalize(f) for f in make_sequence(int(G["seed"]) + b, n, H, W)] for b in range(batch)]
    frames = [torch.cat([s_[t] for s_ in seqs], dim=0).cuda() for t in range(n)]
    gts = _gts(n - 2, H, W, int(G["gt_seed"]), batch)
    for p in model.parameters():
        p.grad = None
    # train / big: the frozen estimator's 1/8-resolution flows come from the fixture (the reference's own), so that the
    # gradient-carrying heads see identical inputs on both sides.  With the build's estimator (EPE ~1e-5 px from the
    # reference at these sizes) enough pre-activations cross a ReLU kink to put isolated 1 % errors on single gradient
    # elements - relative L2 2e-4 instead of 5e-6, measured - which says nothing about the backward kernels.
    small = None
    if any(k.startswith("small/") for k in G):
        small = {tuple(int(v) for v in k[6:].split("_")): torch.from_numpy(G[k]).cuda() for k in G if k.startswith("small/")}
    loss, outs = train.forward_backward(model, frames, gts, small=small)
    assert abs(loss - float(G["loss"])) < 1e-4 * float(G["loss"])
    sub = 4 if case == "c1" else 8
    for k, o in enumerate(outs):
        assert float((o[:, :, ::sub, ::sub].cpu() - torch.from_numpy(G["out%d" % k])).abs().max()) < 1e-3
    params = dict(model.named_parameters())
    names = [str(s) for s in G["names"]]
    assert sorted(names) == sorted(k for k in params if not k.startswith("ofe."))
    assert all(p.grad is None for k, p in params.items() if k.startswith("ofe."))     # the estimator is frozen
    worst, rel_l2 = {}, {}
    for name in names:
        g = params[name].grad
        assert g is not None and g.shape == params[name].shape, name
        g = g.detach().double().reshape(-1).cpu()
        l2 = float(G["l2/" + name])
        rms = l2 / g.numel() ** 0.5
        want = torch.from_numpy(G["val/" + name]).double()
        d = g[::int(G["step/" + name])] - want
        worst[name] = max(float(d.abs().max()) / rms, abs(float(g.norm()) - l2) / l2)
        rel_l2[name] = max(float(d.norm() / want.norm()), abs(float(g.norm()) - l2) / l2)
    if case == "c1":
        bad = {k: v for k, v in worst.items() if v > GRAD_TOL}          # every sampled element within 2e-4 of the RMS
        assert not bad, bad
        return
    # train / big (10 240 / 9 216 coarse pixels per channel instead of 1 024; 10^7 pre-activations in the context encoder):
    # the two fp32 evaluations of the graph put a handful of pre-activations on different sides of a ReLU kink and a few
    # pixels on different sides of getOcc's `mean <= 1.0` threshold (AccFlow_.py:131-134, a 0/1 input of accplus.conv1 for
    # all 256 channels).  Each such pixel changes single gradient elements by ~1/pixels of their value - isolated errors of
    # 1e-3 .. 5e-3 of the RMS on < 10 % of the elements, measured - while the gradient as a whole agrees to 1.5e-4
    # (relative L2 of the sample; 5e-6 at C1 size where no pixel flips).  A wrong backward kernel - the step-mode slices, the
    # batch > 1 paths, the atomic fallback of the deformable convolution's backward beyond 4096-pixel planes - moves every
    # element it touches by far more.  Gate: relative L2 and norm within 5e-4 for every one of the 73 parameters.
    bad = {k: v for k, v in rel_l2.items() if v > 5e-4}
    assert not bad, bad
    assert max(worst.values()) < 2e-2, max(worst.items(), key=lambda kv: kv[1])     # (no gross outlier either)


def test_train_step_lowers_the_loss():
    """AdamW steps on one sequence (train_acc.py:72-87,210-234 without the scheduler): the loss must go down, the
    estimator's parameters must not move, the packs must follow the updated weights (PackCache keys on the version)."""
    from accflow_amd import train
    model, make_sequence, normalize = _setup()
    frames = [normalize(f).cuda() for f in make_sequence(7, 4, 64, 96)]
    gts = _gts(2, 64, 96, 5)
    ofe0 = [p.detach().clone() for p in model.ofe.parameters()]
    opt = torch.optim.AdamW(train.trainable_parameters(model), lr=2e-4, weight_decay=1e-5, eps=1e-8)
    losses = [train.train_step(model, opt, frames, gts)[0] for _ in range(4)]
    assert losses[-1] < losses[0], losses
    assert all(torch.equal(a, b) for a, b in zip(ofe0, model.ofe.parameters()))
    # inference after the update sees the new weights: the eval forward equals the last training forward's successor
    with torch.no_grad():
        out = model(frames)
    l_eval = sum(float((o - g).abs().mean()) for o, g in zip(out, gts))
    l_next, _ = train.forward_backward(model, frames, gts)
    assert abs(l_eval - l_next) < 1e-3 * l_next


def test_train_acc_cli_runs_saves_and_resumes(tmp_path, monkeypatch):
    """The training front end (accflow_amd/train_acc.py) end to end on synthetic sequences: 3 steps, validation,
    latest / numbered / final checkpoints in the reference's format (`module.` keys + .state), then --resume auto."""
    from accflow_amd import train_acc
    monkeypatch.setenv("ACCFLOW_SYNTHETIC", "1")
    monkeypatch.setenv("ACCFLOW_SYNTH_SAMPLES", "4")
    monkeypatch.delenv("ACCFLOW_CVO_LMDB", raising=False)
    cfg = tmp_path / "c.yml"
    cfg.write_text("exp_name: Acc+RAFT-debug\nepochs: 2\nbatch_per_gpu: 2\nimage_size: [64, 64]\nlog_freq: 1\nvalid_freq: 1000\n"
                   "flow_pretrained: none.pth\nlr: !!float 1.2e-4\n")
    n = train_acc.main(["-c", str(cfg), "--steps", "3", "--out", str(tmp_path), "--valid-batches", "1"])
    assert n == 3
    ck = tmp_path / "checkpoints" / "Acc+RAFT-debug"
    names = sorted(p.name for p in ck.iterdir())
    assert names == ["000003.pth", "000003.state", "final.pth", "latest.pth", "latest.state"]
    sd = torch.load(ck / "latest.pth")
    assert all(k.startswith("module.") for k in sd) and "module.accplus.conv2.4.scale" in sd
    st = torch.load(ck / "latest.state")
    assert st["iter"] == 3 and {"optimizer", "scheduler"} <= set(st)
    with pytest.raises(SystemExit):                 # a fresh run never overwrites
        train_acc.main(["-c", str(cfg), "--steps", "1", "--out", str(tmp_path)])
    assert train_acc.main(["-c", str(cfg), "--steps", "4", "--out", str(tmp_path), "--resume", "auto", "--valid-batches", "1"]) == 4
    # a resume from the MIDDLE of an epoch (step 4 of 2 epochs x 3 iterations... here 2 x 4) must run to the
    # scheduled end - skipping the batches the epoch had consumed - and still validate, save and write final.pth there
    (ck / "final.pth").unlink()
    st4 = torch.load(ck / "latest.state")
    assert st4["iter"] == 4 and "best_epe" in st4 and st4["synthetic"] is True
    n_end = train_acc.main(["-c", str(cfg), "--out", str(tmp_path), "--resume", "auto", "--valid-batches", "1"])
    assert n_end == 8                                   # 2 epochs x 4 iterations (8 samples, batch 2)
    assert torch.load(ck / "latest.state")["iter"] == 8 and (ck / "final.pth").exists()
    numbered = [p.name for p in ck.iterdir() if p.name.endswith(".pth") and p.name[:6].isdigit()]
    assert len(numbered) <= 2                           # train_acc.py:296-302: at most two numbered checkpoints beside latest


def test_step_api_and_sequence_schedule_give_the_same_gradients(monkeypatch):
    """train.forward_backward batches what does not depend on the accumulated flow over the steps of a sequence (estimator,
    context encoder, FlowEncoder of flow_ini / dflow, blending masks: one shared tape) and overlaps the backward of step k with
    the forward of step k+1; train.fusion_step_fw alone is the reference's schedule (AccFlow.iter per step: its own
    estimator / context calls, everything on the step's tape).  Same sums in another order: gradients agree to 1e-4 of
    their RMS, with and without the stream overlap."""
    from accflow_amd import backward as B
    from accflow_amd import ops, train
    model, make_sequence, normalize = _setup()
    frames = [normalize(f).cuda() for f in make_sequence(77, 4, 64, 96, batch=2)]
    gts = _gts(2, 64, 96, 9)
    gts = [g.repeat(2, 1, 1, 1) * torch.tensor([1.0, -0.5]).view(2, 1, 1, 1).cuda() for g in gts]
    params = train.trainable_parameters(model)

    def grads(fn):
        for p in params:
            p.grad = None
        fn()
        torch.cuda.synchronize()
        return [p.grad.detach().clone() for p in params]

    def stepwise():
        flow = None
        for k, i in enumerate(range(2, len(frames))):
            t = train.Tape()
            # (mode passed explicitly: a bare call defaults to bf16x6 - nobody reads a range flag here; the inputs are in range)
            small, up = train.fusion_step_fw(t, model, frames[i], frames[i - 1], frames[0], flow, mode=train.TRAIN_CONV_MODE)
            with ops.conv_mode(train.TRAIN_CONV_MODE):
                up.g = B.l1_grad(up.v, gts[k], 1.0 / up.v.numel())
                t.backward()
            flow = small.v
    want = grads(stepwise)
    for batched, overlap in ((True, False), (False, True), (False, False)):
        monkeypatch.setattr(train, "BATCHED_BACKWARD", batched)   # one backward over all steps (the tape's step mode) ...
        monkeypatch.setattr(train, "OVERLAP_BACKWARD", overlap)   # ... or one per step, optionally under the next forward
        got = grads(lambda: train.forward_backward(model, frames, gts))
        for p, a, b in zip(params, got, want):
            rms = float(b.pow(2).mean().sqrt().clamp_min(1e-30))
            assert float((a - b).abs().max()) / rms < 1e-4, (batched, overlap, tuple(p.shape))


def test_graphed_forward_backward_replays_the_eager_step():
    """train.GraphedForwardBackward: forward + backward captured once in a HIP graph.  A replay on NEW inputs, and a replay after
    an optimizer step changed the weights (the trainable packs are rebuilt by kernels inside the graph), give the eager
    step's loss and gradients (float atomics in the weight-gradient kernels: tolerance, not bits)."""
    from accflow_amd import train
    model, make_sequence, normalize = _setup()
    mk = lambda seed: [normalize(f).cuda() for f in make_sequence(seed, 4, 64, 96)]   # noqa: E731
    fa, fb = mk(5), mk(6)
    ga, gb = _gts(2, 64, 96, 1), _gts(2, 64, 96, 2)
    params = train.trainable_parameters(model)
    g = train.GraphedForwardBackward(model, fa, ga)
    opt = torch.optim.SGD(params, lr=1e-3)

    def check(frames, gts):
        loss_g, _ = g(frames, gts)
        got = [p.grad.detach().clone() for p in params]
        for p in params:
            p.grad = None
        loss_e, _ = train.forward_backward(model, frames, gts)
        torch.cuda.synchronize()
        assert abs(loss_g - loss_e) < 1e-5 * abs(loss_e)
        for p, a in zip(params, got):
            rms = float(p.grad.pow(2).mean().sqrt().clamp_min(1e-30))
            assert float((a - p.grad).abs().max()) / rms < 1e-4, tuple(p.shape)
    check(fb, gb)                      # new inputs through the static buffers
    opt.step()                         # the weights move (gradients of the eager pass just made)
    check(fa, ga)                      # the graph re-packs them itself
