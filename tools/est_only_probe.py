"""Probe: time of the estimator half (11 pairs: encoders + correlation + 12 iterations) and of the fusion chain half of one
AccFlow(RAFT) 7 x 480x1024 sequence, each alone in a loop (one stream of work at a time).
    python tools/est_only_probe.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from accflow_amd import ops
    from accflow_amd.data.synthetic import make_sequence, make_state_dict, normalize
    from accflow_amd.networks import build_flow_estimator
    from accflow_amd.networks.AccFlow_ import AccFlow
    model = AccFlow(build_flow_estimator("acc|raft"))
    model.load_state_dict(make_state_dict(model), strict=True)
    model = model.cuda().eval()
    frames = [normalize(f).cuda() for f in make_sequence(1000, 7, 480, 1024)]
    pairs = model.pair_schedule(len(frames))
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    with torch.no_grad(), ops.guard_scope(flag):
        small = model.estimate_small(frames, pairs)
        by_pair = {p: small[k:k + 1] for k, p in enumerate(pairs)}
        t_est = timed(lambda: model.estimate_small(frames, pairs))
        t_chain = timed(lambda: model.fuse_chain(frames, by_pair))
        t_all = timed(lambda: model(images=frames))
    print("estimator alone %.3f ms, fusion chain alone %.3f ms, whole forward %.3f ms per sequence" % (t_est, t_chain, t_all))


if __name__ == "__main__":
    main()
