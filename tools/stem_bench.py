#!/usr/bin/env python3
"""7x7 stride-2 stem (3 -> 64, 480x1024 images): us per launch of the three forms the encoders use.  Run twice:
ACCFLOW_CONV_STEM=0 (im2col kernel) and default (csrc/conv_stem.hip)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from accflow_amd import ops  # noqa: E402


def timeit(fn, reps=20, rounds=3):
    best = 1e9
    for _ in range(rounds):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * s.elapsed_time(e) / reps)
    return best


for B in (7, 6):
    x = torch.randn(B, 3, 480, 1024, device="cuda")
    w = torch.randn(64, 3, 7, 7, device="cuda") * 0.05
    pk = ops.PackedConv(w, torch.randn(64, device="cuda"), stride=2, padding=3)
    out = torch.empty((B, 64, 240, 512), device="cuda")
    o16 = ops.S16.empty(B, 64, 240, 512, "cuda")
    t_relu = timeit(lambda: ops.conv2d(pk, x, act=ops.ACT_RELU, out=out))
    t_stats = timeit(lambda: ops.conv2d(pk, x, out=out, want_stats=True))
    stem = os.environ.get("ACCFLOW_CONV_STEM", "1") != "0"
    t_16 = timeit(lambda: ops.conv2d(pk, x, act=ops.ACT_RELU, out16=o16, fp32_out=False)) if stem else float("nan")
    print("stem=%d B%d: relu->fp32 %.1f us, raw+stats %.1f us, relu->S16 %.1f us" % (stem, B, t_relu, t_stats, t_16))
