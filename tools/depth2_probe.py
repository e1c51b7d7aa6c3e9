"""Experiment: two sequences' estimators in flight on two streams (each covering all 11 pairs per launch) instead of one
sequence split into two pair groups; fusion chains on side streams as in SequencePipeline.  Run on the GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accflow_amd.networks.raft.raft as _raft  # noqa: E402
from accflow_amd import ops  # noqa: E402
from accflow_amd.data.synthetic import make_sequence, make_state_dict, normalize  # noqa: E402
from accflow_amd.networks import build_flow_estimator  # noqa: E402
from accflow_amd.networks.AccFlow_ import AccFlow  # noqa: E402
from accflow_amd.parallel import SequencePipeline  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    model = AccFlow(build_flow_estimator("acc|raft"))
    model.load_state_dict(make_state_dict(model), strict=True)
    model = model.to(dev).eval()
    frames = [normalize(f).to(dev) for f in make_sequence(1000, 7, 480, 1024)]
    ref = [o.clone() for o in model(images=frames)]
    N = frames[0].shape[0]
    pairs = model.pair_schedule(len(frames))

    ES = [torch.cuda.Stream(dev) for _ in range(3)]   # created once: the caching allocator keeps one pool per stream
    CS = [torch.cuda.Stream(dev) for _ in range(3)]

    def run(n, est_streams, group_streams):
        _raft.N_STREAMS = group_streams
        es, cs = ES[:est_streams], CS[:est_streams]
        main_s = torch.cuda.current_stream()
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        pend = []
        outs = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            e, c = es[k % est_streams], cs[k % est_streams]
            e.wait_stream(main_s)
            with ops.guard_scope(flag):
                with torch.cuda.stream(e):
                    small = model.estimate_small(frames, pairs)
                    by = {p: small[i * N:(i + 1) * N] for i, p in enumerate(pairs)}
                c.wait_stream(e)
                with torch.cuda.stream(c):
                    outs = model.fuse_chain(frames, by)
                    ev = torch.cuda.Event(); ev.record(c)
            pend.append((ev, outs, small))
            if len(pend) > est_streams:
                pend.pop(0)[0].synchronize()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e3
        return dt, outs

    for est_streams, group_streams in ((1, 2), (2, 1), (1, 2), (2, 1), (1, 2), (2, 1), (3, 1)):
        run(3, est_streams, group_streams)
        dt, outs = run(12, est_streams, group_streams)
        err = max(float((a - b).abs().max()) for a, b in zip(outs, ref))
        print(f"sequences in flight {est_streams}, pair-group streams {group_streams}: {dt:.2f} ms per sequence (max |diff| vs forward {err:.1e})",
              flush=True)


if __name__ == "__main__":
    main()
