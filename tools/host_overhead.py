"""Host-side cost of issuing one sequence (Python + ctypes + HIP launches) vs the GPU time of the step: the host must
stay ahead of the GPU for the pipeline to hold.  Run on the GPU box: python tools/host_overhead.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accflow_amd.data.synthetic import make_sequence, make_state_dict, normalize  # noqa: E402
from accflow_amd.networks import build_flow_estimator  # noqa: E402
from accflow_amd.networks.AccFlow_ import AccFlow  # noqa: E402
from accflow_amd.parallel import SequencePipeline  # noqa: E402


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--ofe", default="raft")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=1024)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = AccFlow(build_flow_estimator("acc|" + a.ofe))
    model.load_state_dict(make_state_dict(model), strict=True)
    model = model.to(dev).eval()
    frames = [normalize(f).to(dev) for f in make_sequence(1000, 7, a.height, a.width)]
    pipe = SequencePipeline(model)
    for _ in range(3):
        pipe.submit(frames)
    pipe.flush()
    torch.cuda.synchronize()
    n = 10
    host = []
    t0 = time.perf_counter()
    for _ in range(n):
        h0 = time.perf_counter()
        p = pipe._launch(frames)          # issue only: no harvest, no wait
        host.append(time.perf_counter() - h0)
        if pipe.pending is not None:
            pipe._harvest(pipe.pending)
        pipe.pending = p
    pipe.flush()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    host.sort()
    print("host issue time per sequence: median %.2f ms (min %.2f, max %.2f); wall per sequence %.2f ms -> host busy %.0f %%"
          % (1e3 * host[n // 2], 1e3 * host[0], 1e3 * host[-1], 1e3 * wall, 100 * host[n // 2] / wall))


if __name__ == "__main__":
    main()
