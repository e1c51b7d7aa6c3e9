"""Per-shape table of the weight-gradient GEMM launches of one eager training step (reference configuration
configs/AccRAFT-CVO.yml: 7 x 256 x 256, batch 6): HIP events around every accflow_conv_wgrad_f32 launch.

    python tools/wgrad_shapes.py [--out profiles/rNN_wgrad_shapes.txt]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from accflow_amd import profiler, train
    from accflow_amd.data.synthetic import make_sequence, make_state_dict, normalize
    from accflow_amd.networks import build_flow_estimator
    from accflow_amd.networks.AccFlow_ import AccFlow
    model = AccFlow(build_flow_estimator("acc|raft"))
    model.load_state_dict(make_state_dict(model), strict=True)
    model = model.cuda().eval()
    frames = [normalize(f).cuda() for f in make_sequence(11, 7, 256, 256, batch=6)]
    g = torch.Generator().manual_seed(3)
    gts = [(3.0 * torch.randn(6, 2, 256, 256, generator=g)).cuda() for _ in range(5)]
    opt = torch.optim.AdamW(train.trainable_parameters(model), lr=1.2e-4, weight_decay=1e-5, eps=1e-8)
    for _ in range(2):
        train.train_step(model, opt, frames, gts)
    wt = profiler.KernelTimer(["conv_wgrad"])
    profiler.ACTIVE = wt
    try:
        for _ in range(3):
            train.train_step(model, opt, frames, gts)
        torch.cuda.synchronize()
    finally:
        profiler.ACTIVE = None
    rows = sorted(wt.by_detail("conv_wgrad").items(), key=lambda kv: -kv[1]["total_ms"])
    lines = ["weight-gradient GEMM launches of 3 eager training steps (7 x 256 x 256, batch 6), per step:"]
    tot = 0.0
    for name, d in rows:
        ms = d["total_ms"] / 3.0
        tot += ms
        lines.append("%-44s launches/step %5.1f  ms/step %7.3f  us/launch %7.1f  TFLOP/s %6.1f" % (
            name, d["launches"] / 3.0, ms, 1e3 * d["total_ms"] / d["launches"], d["work"] / (d["total_ms"] * 1e-3) / 1e12))
    lines.append("total %.3f ms per step" % tot)
    text = "\n".join(lines)
    print(text)
    if a.out:
        open(a.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
