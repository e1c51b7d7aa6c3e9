#!/usr/bin/env python3
"""Warm-start mode: how the long-range estimate converges with the number of refinement iterations when
it starts from the composed accumulated flow instead of zero.  Prints, for k warm iterations, the EPE of
AccFlow(warm_start=True, warm_iters=k) against the cold 12-iteration result, and against the analytic ground truth of
the synthetic sequence.  With the offline random-init weights the estimator is not a trained contraction, so the
"iterations to equal EPE" figure proper needs released checkpoints (pass --acc_ckpt); the table still shows what the
mode costs and returns on this machine."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from accflow_amd.data.synthetic import gt_flow, make_sequence, make_state_dict, normalize  # noqa: E402
from accflow_amd.networks import build_flow_estimator  # noqa: E402
from accflow_amd.networks.AccFlow_ import AccFlow  # noqa: E402


def epe(a, b):
    return float(torch.norm(a - b, p=2, dim=1).mean())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=7)
    ap.add_argument("--acc_ckpt", default=None)
    a = ap.parse_args()
    model = AccFlow(build_flow_estimator("acc|raft"))
    sd = torch.load(a.acc_ckpt, map_location="cpu") if a.acc_ckpt else make_state_dict(model)
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    frames = [normalize(f).cuda() for f in make_sequence(1000, a.frames, a.height, a.width)]
    gt = gt_flow(a.frames - 1, 0, a.height, a.width)[None].cuda()

    def run(**kw):
        for k, v in kw.items():
            setattr(model, k, v)
        model(images=frames)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model(images=frames)[-1]
        torch.cuda.synchronize()
        return out, 1e3 * (time.perf_counter() - t0)

    cold, t_cold = run(warm_start=False)
    print("cold schedule, 12 iterations: %.1f ms per sequence, EPE vs analytic flow %.3f px" % (t_cold, epe(cold, gt)))
    for k in (1, 2, 3, 4, 6, 8, 12):
        out, t = run(warm_start=True, warm_iters=k)
        print("warm start, %2d iterations on the long-range pairs: %.1f ms per sequence, EPE vs cold %.4f px, vs analytic flow %.3f px"
              % (k, t, epe(out, cold), epe(out, gt)))
