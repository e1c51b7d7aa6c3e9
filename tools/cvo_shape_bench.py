"""Throughput at the shape test_cvo.py drives (test_cvo.py:114-116): batches of 10 CVO sequences, 7 frames of 512x512,
12 GRU iterations = 110 estimator pairs per launch, through the sequence pipeline.  Run on the GPU box."""
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
    dev = torch.device("cuda", 0)
    model = AccFlow(build_flow_estimator("acc|raft"))
    model.load_state_dict(make_state_dict(model), strict=True)
    model = model.to(dev).eval()
    for N in (10, 1):
        frames = [normalize(f).to(dev) for f in make_sequence(1300, 7, 512, 512, batch=N)]
        pipe = SequencePipeline(model)
        for _ in range(2):
            pipe.submit(frames)
        pipe.flush()
        torch.cuda.synchronize()
        n = 6
        t0 = time.perf_counter()
        for _ in range(n):
            pipe.submit(frames)
        pipe.flush()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"batch of {N} sequences (7 x 512x512, 12 iterations): {1e3 * dt:.1f} ms per batch = {1e3 * dt / N:.2f} ms per sequence "
              f"= {11 * N / dt:.0f} frame-pairs/s", flush=True)


if __name__ == "__main__":
    main()
