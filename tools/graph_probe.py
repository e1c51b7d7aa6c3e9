"""Can one RAFT pair evaluation (configs[1]: 480x1024, 12 iterations, batch 1 - ~400 kernel launches in ~6 ms, i.e.
launch-bound) be captured in a HIP graph through torch.cuda.graph?  Run on the GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accflow_amd import ops  # noqa: E402
from accflow_amd.data.synthetic import make_sequence, make_state_dict, normalize  # noqa: E402
from accflow_amd.networks import build_flow_estimator  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    m = build_flow_estimator("raft")
    m.load_state_dict(make_state_dict(m), strict=True)
    m = m.to(dev).eval()
    fr = [normalize(f).to(dev) for f in make_sequence(1000, 2, 480, 1024)]
    ref = m(fr[1], fr[0], iters=12).clone()
    torch.cuda.synchronize()

    def timed(fn, n=10):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    print("eager: %.3f ms per pair" % timed(lambda: m(fr[1], fr[0], iters=12)))
    a, b = fr[1].clone(), fr[0].clone()
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), ops.guard_scope(flag):
        for _ in range(2):
            m(a, b, iters=12)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with ops.guard_scope(flag):
        with torch.cuda.graph(g):
            out = m(a, b, iters=12)
    torch.cuda.synchronize()
    a.copy_(fr[1]); b.copy_(fr[0])
    g.replay()
    torch.cuda.synchronize()
    print("graph replay max |diff| vs eager:", float((out - ref).abs().max()), "flag", int(flag.item()))
    print("graph: %.3f ms per pair" % timed(lambda: g.replay()))


if __name__ == "__main__":
    main()
