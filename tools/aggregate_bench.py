"""Time the GMA aggregation GEMM (ops.gma_aggregate_t) at the C5 size: P = 90*160 pixels, D = 128 rows per item.
Run on the GPU box:  python tools/aggregate_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accflow_amd import ops  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    h, w, D = 90, 160, 128
    P = h * w
    g = torch.Generator().manual_seed(3)
    attn = torch.softmax(torch.randn(P, P, generator=g) * 3, dim=0).cuda()[None].contiguous()  # j-major columns sum to 1
    gamma = torch.tensor([0.5]).cuda()
    qk = torch.randn(3, 2 * D, h, w, generator=g).cuda()
    ms = timed(lambda: ops.gma_attention_t(qk, D, D ** -0.5), reps=3)
    print(f"attention build (q.k GEMM + column softmax), 3 items: {ms:.3f} ms = {ms / 3:.3f} ms per item "
          f"({3 * P * P * 4 / ms / 1e6:.0f} GB/s of final matrix)", flush=True)
    for n in (1, 2, 3):
        v = torch.randn(1, n * D, h, w, generator=g).cuda()
        fm = torch.randn(1, n * D, h, w, generator=g).cuda()
        for mode, name in ((ops.CONV_F16X3, "f16x3"), (ops.CONV_BF16X6, "bf16x6")):
            ms = timed(lambda: ops.gma_aggregate_t(attn, v, fm, gamma, mode=mode))
            flop = 2.0 * n * D * P * P
            print(f"items {n} mode {name}: {ms:.3f} ms  {flop / ms / 1e9:.1f} TFLOP/s (fp32-equivalent)  "
                  f"attention read {P * P * 4 / ms / 1e6:.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
