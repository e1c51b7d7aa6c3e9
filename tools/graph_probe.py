"""Probe: AccFlow(RAFT) 7 x 480x1024 forward captured in ONE HIP graph vs the eager forward, one sequence at a time.
    python tools/graph_probe.py [--steps 20]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=1024)
    a = ap.parse_args()
    from accflow_amd import ops
    from accflow_amd.data.synthetic import make_sequence, make_state_dict, normalize
    from accflow_amd.networks import build_flow_estimator
    from accflow_amd.networks.AccFlow_ import AccFlow
    model = AccFlow(build_flow_estimator("acc|raft"))
    model.load_state_dict(make_state_dict(model), strict=True)
    model = model.cuda().eval()
    frames = [normalize(f).cuda() for f in make_sequence(1000, 7, a.height, a.width)]
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")

    def eager():
        with torch.no_grad(), ops.guard_scope(flag):
            return model(images=frames)

    for _ in range(3):
        ref = eager()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eager()
    torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / a.steps
    ref = [o.clone() for o in eager()]
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        with torch.no_grad(), ops.guard_scope(flag):
            outs = model(images=frames)
    keep = list(getattr(ops._tls, "ksplit_ws", {}).values())
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        g.replay()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / a.steps
    err = max(float((x - y).abs().max()) for x, y in zip(outs, ref))
    print("eager %.3f ms  graph %.3f ms per sequence  (max |diff| %.3g, flag %d)" % (1e3 * te, 1e3 * tg, err, int(flag.item())))


if __name__ == "__main__":
    main()
