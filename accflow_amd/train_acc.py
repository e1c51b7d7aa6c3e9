#!/usr/bin/env python3
"""Training front end with the reference's train_acc.py semantics on the HIP path.

    python -m accflow_amd.train_acc -c configs/AccRAFT-CVO.yml [--steps N] [--resume auto|STEP] [--out DIR]

Reads the reference's YAML options unchanged (configs/*.yml: exp_name, epochs, lr, wdecay, epsilon, batch_per_gpu, clip,
add_noise, log_freq, valid_freq, image_size, flow_pretrained ...) and mirrors train_acc.py:113-312: frozen estimator
loaded from `flow_pretrained`, AdamW + linear one-cycle schedule (:72-87), the noise augmentation (:216-220),
sequence_loss_acc (loss.py:30-45), gradient clipping, periodic validation on the CVO test split with best-EPE
checkpoint rotation (:253-307), latest / numbered / final checkpoints as `.pth` (weights, keys prefixed `module.` like the
reference's nn.DataParallel saves) + `.state` (iter, optimizer, scheduler) (:96-110).

Differences, all deliberate:
  * one process per GPU (RANK / WORLD_SIZE / LOCAL_RANK from torchrun) with ONE gradient all-reduce per step
    (train.allreduce_grads) instead of nn.DataParallel; `batch_per_gpu` is the per-rank batch, `gpus` only sizes the step
    count when WORLD_SIZE is unset;
  * `mixed_precision` is read and ignored: forward and backward run in fp32-equivalent arithmetic (the heads' forward on
    the fp16 hi + lo split with its range guard, a tripped step redone in bf16x6: train.TRAIN_CONV_MODE), there is no
    GradScaler;
  * the frozen estimator stays in eval() (its BatchNorm uses the running statistics): train_acc.py:169's model.train()
    also flips the frozen RAFT's BatchNorm to batch statistics and lets its running averages drift, a side effect;
  * existing log / checkpoint directories are never renamed (train_acc.py:39-43 archives them): a fresh run refuses to
    overwrite, --resume continues;
  * a run ENDS at `epochs * iters_per_epoch` optimizer steps, also after --resume from the middle of an epoch: the resumed
    epoch skips the batches it had already seen (the sampler is seeded per epoch) and the loop stops at the step count the
    one-cycle schedule was built for.  (The reference restarts the whole epoch and runs past the schedule's
    `total_steps`, where OneCycleLR raises.)  The final step is always validated and saved;
  * `best_epe` / `best_step` travel in the `.state` file (absent in reference states: the first validation then counts as
    the best, as in the reference);
  * without the CVO training LMDB or the `flow_pretrained` checkpoint the run STOPS unless --synthetic (or
    ACCFLOW_SYNTHETIC=1) asks for the synthetic sequences / the deterministic synthetic estimator weights - a smoke run
    can then not be mistaken for a real one: its checkpoints carry `"synthetic": true` in the `.state` file.
"""
import argparse
import logging
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DEFAULTS = dict(epochs=60, lr=1.2e-4, wdecay=1e-5, epsilon=1e-8, batch_per_gpu=6, clip=1.0, add_noise=True, log_freq=100,
                valid_freq=1000, image_size=[256, 256], flow_pretrained=None, valid_sample=500, gpus=[0], loss_type="L1")


def parse_options(path):
    import yaml
    with open(path) as f:
        opt = yaml.safe_load(f) or {}
    for k, v in DEFAULTS.items():
        opt.setdefault(k, v)
    if "exp_name" not in opt:
        raise ValueError("%s: no exp_name" % path)
    if str(opt["loss_type"]).upper() != "L1":
        raise NotImplementedError("loss_type %r: train_acc.py only implements the L1 sequence loss" % opt["loss_type"])
    return argparse.Namespace(**opt)


def sequence_loss_metrics(flow_preds, flow_gts):
    """The metrics half of loss.py:37-45 (the loss half and its gradient are train.forward_backward)."""
    epe = torch.sum((flow_preds[-1] - flow_gts[-1]) ** 2, dim=1).sqrt().view(-1)
    return {"epe": epe.mean().item(), "1px": (epe < 1).float().mean().item(), "3px": (epe < 3).float().mean().item(),
            "5px": (epe < 5).float().mean().item()}


def fetch_optimizer(args, params, num_steps):
    """train_acc.py:72-87"""
    opt = torch.optim.AdamW(params, lr=args.lr, weight_decay=args.wdecay, eps=args.epsilon)
    sch = torch.optim.lr_scheduler.OneCycleLR(optimizer=opt, max_lr=args.lr, total_steps=num_steps + 100, pct_start=0.05,
                                              cycle_momentum=False, anneal_strategy="linear")
    return opt, sch


def save_ckpt(step, scheduler, optimizer, model, ckpt_dir, latest=True, extra=None):
    """train_acc.py:96-110; weights under the `module.` prefix nn.DataParallel gives the reference's checkpoints."""
    stem = "latest" if latest else "%06d" % step
    sd = {"module." + k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.save(sd, os.path.join(ckpt_dir, stem + ".pth"))
    state = {"iter": step, "scheduler": scheduler.state_dict(), "optimizer": optimizer.state_dict()}
    state.update(extra or {})
    torch.save(state, os.path.join(ckpt_dir, stem + ".state"))


def rotate_ckpts(ckpt_dir):
    """train_acc.py:296-302: once four or more .pth files exist (latest.pth counts), the oldest NUMBERED one goes - at most
    two numbered checkpoints remain beside latest."""
    while True:
        pths = [x for x in os.listdir(ckpt_dir) if x.endswith(".pth") and x != "final.pth"]
        numbered = sorted(x for x in pths if x[:6].isdigit())
        if len(pths) < 4 or not numbered:
            return
        old = numbered[0]
        os.remove(os.path.join(ckpt_dir, old))
        st = os.path.join(ckpt_dir, old[:-4] + ".state")
        if os.path.exists(st):
            os.remove(st)


def add_noise(images):
    """train_acc.py:216-220 (one noise field added to every frame)."""
    stdv = np.random.uniform(0.0, 5.0)
    noise = stdv * torch.randn(*images[0].shape, device=images[0].device)
    noise = 2 * (torch.clamp(noise, 0.0, 255.0) / 255.0) - 1
    return [x + noise for x in images]


def validate(model, loader, dev, limit):
    """train_acc.py:253-271 on this rank: mean of sequence_loss_acc's metrics over the validation batches."""
    from accflow_amd.eval_cvo import preprocess
    mets, last = [], None
    with torch.no_grad():
        for i, batch in enumerate(loader):
            if limit is not None and i >= limit:
                break
            d = preprocess(batch, dev)
            out = model(d["imgs"])
            gts = d["bflows"][:len(out)]
            m = sequence_loss_metrics(out, gts)
            m["loss"] = sum(float((o - g).abs().mean()) for o, g in zip(out, gts))
            mets.append(m)
            last = out[-1]
    if not mets:
        raise RuntimeError("validate: the validation loader yielded no batch")
    return {"val_" + k: sum(m[k] for m in mets) / len(mets) for k in mets[0]}, last


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", "-c", type=str, required=True)
    ap.add_argument("--steps", type=int, default=None, help="stop after this many optimizer steps (smoke runs)")
    ap.add_argument("--resume", type=str, default=None, help="'auto' (latest) or a saved step number (train_acc.py:27-32)")
    ap.add_argument("--out", type=str, default=".", help="root of logs/<exp_name> and checkpoints/<exp_name>")
    ap.add_argument("--valid-batches", type=int, default=None, help="cap on validation batches per validation")
    ap.add_argument("--synthetic", action="store_true",
                    help="smoke run: synthetic sequences / synthetic estimator weights where the real ones are missing")
    a = ap.parse_args(argv)
    args = parse_options(a.config)
    synthetic_ok = a.synthetic or os.environ.get("ACCFLOW_SYNTHETIC", "0") == "1"
    import torch.distributed as dist
    from accflow_amd import train
    from accflow_amd.data.dataset import fetch_train_dataloader, fetch_valid_dataloader
    from accflow_amd.eval_cvo import _strip_module, preprocess
    from accflow_amd.networks import build_flow_estimator
    from accflow_amd.networks.AccFlow_ import AccFlow
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    log_dir, ckpt_dir = os.path.join(a.out, "logs", args.exp_name), os.path.join(a.out, "checkpoints", args.exp_name)
    # Every go / no-go decision that depends on the (shared) file system is taken by EVERY rank before the process group
    # exists: a rank that stops alone would leave the others blocked in their first collective.
    if a.resume is None and os.path.isdir(ckpt_dir) and os.listdir(ckpt_dir):
        raise SystemExit("%s holds checkpoints: pass --resume auto or another --out" % ckpt_dir)
    if not synthetic_ok:
        from accflow_amd.data.dataset import find_cvo_lmdb
        missing = [what for what, ok in (("the CVO training LMDB (ACCFLOW_CVO_LMDB / data/datasets/CVO_full/cvo_train.lmdb)",
                                          find_cvo_lmdb(True) is not None),
                                         ("flow_pretrained %r" % args.flow_pretrained,
                                          bool(args.flow_pretrained) and os.path.isfile(args.flow_pretrained))) if not ok]
        if missing:
            raise SystemExit("train_acc: %s not found - a real run needs them; pass --synthetic (or ACCFLOW_SYNTHETIC=1) for a "
                             "smoke run on synthetic sequences / synthetic estimator weights" % " and ".join(missing))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    if rank == 0:
        os.makedirs(log_dir, exist_ok=True)
        os.makedirs(ckpt_dir, exist_ok=True)
    if world > 1:
        dist.barrier()
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING, format="%(asctime)s %(message)s",
                        handlers=[logging.StreamHandler()] + ([logging.FileHandler(os.path.join(log_dir, "base_%s.log" % args.exp_name))]
                                                              if rank == 0 else []))
    log = logging.getLogger("base")

    # ---- data (train_acc.py:127-152) ----
    crop = args.image_size[0] if isinstance(args.image_size, (list, tuple)) else args.image_size
    loader, dst = fetch_train_dataloader(keys=["bflows"], batch=args.batch_per_gpu, crop_size=crop, split="clean+final",
                                         workers=0, rank=rank, world=world, seed=1234)
    vloader, _ = fetch_valid_dataloader(keys=["bflows"], split="clean", batch=args.batch_per_gpu)
    per_epoch = max(1, len(loader))
    num_steps = per_epoch * args.epochs
    log.info("Train on %d samples with batch %d x %d ranks, %d iters/epoch, %d iters in total", len(dst), args.batch_per_gpu,
             world, per_epoch, num_steps)

    # ---- model & optimizer (:155-173) ----
    ofe = build_flow_estimator(args.exp_name)
    from accflow_amd.data.dataset import find_cvo_lmdb as _find
    synthetic_used = _find(True) is None
    if args.flow_pretrained and os.path.isfile(args.flow_pretrained):
        ofe.load_state_dict(_strip_module(torch.load(args.flow_pretrained, map_location="cpu")))
    else:
        from accflow_amd.data.synthetic import make_state_dict
        log.warning("SMOKE RUN: flow_pretrained %r not found, the estimator gets the build's deterministic synthetic weights",
                    args.flow_pretrained)
        ofe.load_state_dict(make_state_dict(ofe), strict=True)
        synthetic_used = True
    for p in ofe.parameters():
        p.requires_grad = False
    model = AccFlow(ofe).to(dev).eval()      # the tape differentiates explicitly; module modes only matter to the estimator
    params = train.trainable_parameters(model)
    log.info("model: %s  trainable %d, frozen %d parameters", args.exp_name, sum(p.numel() for p in params),
             sum(p.numel() for p in ofe.parameters()))
    optimizer, scheduler = fetch_optimizer(args, params, num_steps)
    step, best_epe, best_step = 0, 1e10, 0
    if a.resume is not None:
        stem = "latest" if a.resume.lower() == "auto" else "%06d" % int(a.resume)
        model.load_state_dict(_strip_module(torch.load(os.path.join(ckpt_dir, stem + ".pth"), map_location="cpu")), strict=True)
        state = torch.load(os.path.join(ckpt_dir, stem + ".state"), map_location="cpu")
        optimizer.load_state_dict(state["optimizer"])
        scheduler.load_state_dict(state["scheduler"])
        step = state["iter"]
        best_epe, best_step = state.get("best_epe", 1e10), state.get("best_step", step)   # (absent in reference states)
        log.info("resumed %s at iter %d", stem, step)
    elif world > 1:                           # every rank starts from rank 0's heads
        for p in model.state_dict().values():
            dist.broadcast(p, 0)

    # forward + backward replayed from a HIP graph (train.GraphedForwardBackward): the loader serves fixed shapes (fixed crop,
    # drop_last), so it is captured once; ACCFLOW_TRAIN_GRAPH=0 runs every step eagerly
    use_graph, graphed = os.environ.get("ACCFLOW_TRAIN_GRAPH", "1") == "1", None
    losses, epes, t_last = [], [], time.time()
    done = step >= num_steps
    for epoch in range(step // per_epoch, args.epochs):
        if done:
            break
        # (a resumed epoch continues behind the batches it had already consumed: same sampler seed, same order; the skip
        # happens at the sampler's index level, no skipped record is read or decoded)
        skip = step - epoch * per_epoch if epoch == step // per_epoch else 0
        loader.sampler.set_epoch(epoch, skip=skip * args.batch_per_gpu)
        for batch in loader:
            step += 1
            d = preprocess(batch, dev)
            images, label = d["imgs"], d["bflows"]
            if args.add_noise:
                images = add_noise(images)
            gts = label[:len(images) - 2]
            if use_graph and (graphed is None or not graphed.shapes_match(images, gts)):
                graphed = train.GraphedForwardBackward(model, images, gts)     # (fixed crop + drop_last: captured once)
            loss, outs = train.train_step(model, optimizer, images, gts, clip=args.clip, scheduler=scheduler,
                                          graphed=graphed if use_graph else None)
            losses.append(loss)
            epes.append(sequence_loss_metrics(outs, label[:len(outs)])["epe"])
            if step % args.log_freq == 0 or step < 25:
                dt = (time.time() - t_last) / len(losses)
                log.info("<epoch:%2d, iter:%6d, t:%.2fs, eta:%.2fh, loss:%.3f, epe:%.3f>", epoch, step, dt,
                         dt * (num_steps - step) / 3600, sum(losses) / len(losses), sum(epes) / len(epes))
                losses, epes, t_last = [], [], time.time()
            # the run ends at the step count the one-cycle schedule was built for (or at --steps)
            last = (a.steps is not None and step >= a.steps) or step >= num_steps
            if step % args.valid_freq == 0 or last:
                if rank == 0:
                    vm, _ = validate(model, vloader, dev, a.valid_batches)
                    if vm["val_epe"] <= best_epe:
                        best_epe, best_step, new_best = vm["val_epe"], step, True
                    else:
                        new_best = False
                    extra = {"best_epe": best_epe, "best_step": best_step, "synthetic": bool(synthetic_used)}
                    save_ckpt(step, scheduler, optimizer, model, ckpt_dir, True, extra)
                    if new_best:
                        save_ckpt(step, scheduler, optimizer, model, ckpt_dir, False, extra)
                        rotate_ckpts(ckpt_dir)
                    log.info("Validation EPE: %.3f, current best EPE: %.3f(step: %s)", vm["val_epe"], best_epe, best_step)
                if world > 1:
                    dist.barrier()
            if last:
                done = True
                break
        if done:
            break
    if rank == 0:
        torch.save({"module." + k: v.detach().cpu() for k, v in model.state_dict().items()}, os.path.join(ckpt_dir, "final.pth"))
        log.info("Finish training")
    if world > 1:
        dist.destroy_process_group()
    return step


if __name__ == "__main__":
    main()
