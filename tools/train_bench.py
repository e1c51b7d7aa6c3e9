"""Time of one training step (accflow_amd/train.py) on the reference's training configuration
(configs/AccRAFT-CVO.yml: 7 frames of 256 x 256, batch_per_gpu 6, AdamW, clip 1.0), synthetic data, one GPU.

    python tools/train_bench.py [--batch 6] [--steps 5] [--warmup 2] [--out profiles/r04_train_bench.json]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=6)
    ap.add_argument("--frames", type=int, default=7)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--out", default=None)
    ap.add_argument("--graph", type=int, default=1, help="1: also time the step with forward + backward replayed from a HIP graph")
    a = ap.parse_args()
    from accflow_amd import profiler, train
    from accflow_amd.data.synthetic import make_sequence, make_state_dict, normalize
    from accflow_amd.networks import build_flow_estimator
    from accflow_amd.networks.AccFlow_ import AccFlow
    model = AccFlow(build_flow_estimator("acc|raft"))
    model.load_state_dict(make_state_dict(model), strict=True)
    model = model.cuda().eval()
    frames = [normalize(f).cuda() for f in make_sequence(11, a.frames, a.size, a.size, batch=a.batch)]
    g = torch.Generator().manual_seed(3)
    gts = [(3.0 * torch.randn(a.batch, 2, a.size, a.size, generator=g)).cuda() for _ in range(a.frames - 2)]
    opt = torch.optim.AdamW(train.trainable_parameters(model), lr=1.2e-4, weight_decay=1e-5, eps=1e-8)
    losses = []
    for _ in range(a.warmup):
        losses.append(train.train_step(model, opt, frames, gts)[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses.append(train.train_step(model, opt, frames, gts)[0])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    ms_graph = None
    if a.graph:
        gfb = train.GraphedForwardBackward(model, frames, gts)
        for _ in range(a.warmup):
            losses.append(train.train_step(model, opt, frames, gts, graphed=gfb)[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            losses.append(train.train_step(model, opt, frames, gts, graphed=gfb)[0])
        torch.cuda.synchronize()
        ms_graph = (time.perf_counter() - t0) * 1e3 / a.steps
    # forward-only (inference path, same batch) for the ratio
    with torch.no_grad():
        model(frames)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            model(frames)
        torch.cuda.synchronize()
    fwd = (time.perf_counter() - t0) * 1e3 / a.steps
    res = {"workload": "train step AccFlow(RAFT) %dx%dx%d batch %d (configs/AccRAFT-CVO.yml)" % (a.frames, a.size, a.size, a.batch),
           "ms_per_train_step": round(ms, 2), "sequences_per_s": round(a.batch / ms * 1e3, 2),
           "ms_per_train_step_graph": None if ms_graph is None else round(ms_graph, 2),
           "sequences_per_s_graph": None if ms_graph is None else round(a.batch / ms_graph * 1e3, 2),
           "ms_inference_forward_same_batch": round(fwd, 2), "losses": [round(x, 4) for x in losses],
           "steps": a.steps, "warmup": a.warmup, "conv_mode_train": train.TRAIN_CONV_MODE}
    print(json.dumps(res))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
