#!/usr/bin/env python3
"""A/B of the direct conv kernel's two input forms on the C3 workload's shapes, interleaved rounds in one process:
fp32 activations (gather + fp16 split in the patch loader) vs pre-split S16 activations (LDS-DMA loader), each with
fp32 and with S16 output.   usage: python tools/s16_conv_bench.py [--reps 20] [--rounds 3]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from accflow_amd import ops  # noqa: E402

SHAPES = [
    ((128, 128), 256, 1, 5, 11, 60, 128),    # GRU zr (two sources)
    ((128, 128), 128, 5, 1, 11, 60, 128),    # GRU q
    ((256,), 192, 3, 3, 11, 60, 128),        # convc2
    ((128,), 256, 3, 3, 11, 60, 128),        # flow head / mask head conv1
    ((256,), 126, 3, 3, 11, 60, 128),        # motion encoder conv
    ((128,), 64, 3, 3, 11, 60, 128),         # convf2
    ((352,), 256, 1, 1, 11, 60, 128),        # convc1 over the S16 lookup
    ((16,), 128, 1, 7, 11, 60, 128),         # convf1 as 1x7
    ((256,), 18, 1, 1, 11, 60, 128),         # flow head conv2: all taps as a 1x1
    ((64,), 64, 3, 3, 7, 240, 512),          # encoder layer1
    ((96,), 96, 3, 3, 7, 120, 256),          # encoder layer2
    ((128,), 128, 3, 3, 7, 60, 128),         # encoder layer3
    ((256,), 128, 3, 3, 1, 60, 128),         # fusion chain, batch 1
    ((512,), 256, 3, 3, 1, 60, 128),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    print("%-44s %10s %10s %10s %10s   (us per launch, min over %d rounds; TFLOP/s of the best S16 form)" % (
        "shape", "f32->f32", "S16->f32", "f32->S16", "S16->S16", a.rounds))
    for cins, Cout, KH, KW, B, H, W in SHAPES:
        Cin = sum(cins)
        xs = [torch.randn(B, c, H, W, device="cuda") for c in cins]
        w = torch.randn(Cout, Cin, KH, KW, device="cuda") * 0.05
        pk = ops.PackedConv(w, torch.randn(Cout, device="cuda"), padding=(KH // 2, KW // 2), C0=cins[0])
        x16 = [ops.to_s16(x) for x in xs]
        out = torch.empty((B, Cout, H, W), device="cuda")
        o16 = ops.S16.empty(B, Cout, H, W, out.device)
        second = lambda l: l[1] if len(l) > 1 else None  # noqa: E731
        variants = {
            "f32->f32": lambda: ops.conv2d(pk, xs[0], second(xs), out=out, act=ops.ACT_RELU),
            "S16->f32": lambda: ops.conv2d(pk, x16[0], second(x16), out=out, act=ops.ACT_RELU),
            "f32->S16": lambda: ops.conv2d(pk, xs[0], second(xs), out16=o16, act=ops.ACT_RELU, fp32_out=False),
            "S16->S16": lambda: ops.conv2d(pk, x16[0], second(x16), out16=o16, act=ops.ACT_RELU, fp32_out=False),
        }
        best = {k: 1e9 for k in variants}
        for _ in range(a.rounds):
            for k, fn in variants.items():
                fn()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(a.reps):
                    fn()
                e.record()
                torch.cuda.synchronize()
                best[k] = min(best[k], 1e3 * s.elapsed_time(e) / a.reps)
        fl = 2.0 * Cin * KH * KW * Cout * B * H * W
        print("%-44s %10.1f %10.1f %10.1f %10.1f   %7.1f TFLOP/s" % (
            "Cin%s Cout%d k%dx%d B%d %dx%d" % ("+".join(map(str, cins)), Cout, KH, KW, B, H, W),
            best["f32->f32"], best["S16->f32"], best["f32->S16"], best["S16->S16"],
            fl / min(best["S16->f32"], best["S16->S16"]) / 1e6), flush=True)


if __name__ == "__main__":
    main()
