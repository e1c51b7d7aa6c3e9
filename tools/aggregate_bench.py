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
    # the product path (S16 mode): pre-split attention, ONE GEMM per run of items sharing it (modules.Aggregate._aggregate16)
    a16 = ops.gma_attention_s16(qk[:1].contiguous(), D, D ** -0.5)
    for n in (1, 2):
        big = torch.randn(n, 3 * D, h, w, generator=g).cuda()
        fmap, out = big[:, :D], big[:, D:2 * D]
        v = torch.randn(n, D, h, w, generator=g).cuda()
        out16 = ops.S16.empty(n, 2 * D, h, w, v.device)
        ms = timed(lambda: ops.gma_aggregate_s16(a16.ptr(), v, fmap[0].data_ptr(), big.stride(0), gamma, out[0].data_ptr(),
                                                 big.stride(0), out16.channels(D, 2 * D).ptr(), out16.bs, n, D, h, w), reps=20)
        print(f"S16 aggregation, {n} item(s) on one attention matrix (pack + GEMM + split-K reduce): {ms * 1e3:.1f} us  "
              f"{2.0 * n * D * P * P / ms / 1e9:.1f} TFLOP/s  attention read {P * P * 4 / ms / 1e6:.0f} GB/s", flush=True)
    if "--s16-only" in sys.argv:
        return
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
