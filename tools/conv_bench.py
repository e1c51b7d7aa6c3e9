#!/usr/bin/env python3
"""Micro-benchmark of accflow_conv2d_f32 on the conv shapes of the C3 workload (GPU only).
usage: python tools/conv_bench.py [--reps 20] [--shapes big|all|<Cin,Cout,KH,KW,stride,B,H,W>...]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from accflow_amd import ops  # noqa: E402

BIG = [
    (384, 256, 1, 5, 1, 11, 60, 128),
    (256, 192, 3, 3, 1, 11, 60, 128),
    (128, 256, 3, 3, 1, 11, 60, 128),
    (256, 126, 3, 3, 1, 11, 60, 128),
    (384, 128, 5, 1, 1, 11, 60, 128),
    (64, 64, 3, 3, 1, 7, 240, 512),
    (96, 96, 3, 3, 1, 7, 120, 256),
    (324, 256, 1, 1, 1, 11, 60, 128),
]
SMALL = [
    (256, 128, 3, 3, 1, 1, 60, 128),
    (257, 256, 3, 3, 1, 1, 60, 128),
    (512, 256, 3, 3, 1, 1, 60, 128),
    (256, 2, 3, 3, 1, 11, 60, 128),
    (256, 2, 3, 3, 1, 1, 60, 128),
    (128, 27, 3, 3, 1, 1, 60, 128),
    (2, 128, 7, 7, 1, 11, 60, 128),
    (3, 64, 7, 7, 2, 7, 480, 1024),
]


def run(shape, reps):
    Cin, Cout, KH, KW, st, B, H, W = shape
    x = torch.randn(B, Cin, H, W, device="cuda")
    w = torch.randn(Cout, Cin, KH, KW, device="cuda") * 0.05
    b = torch.randn(Cout, device="cuda")
    pk = ops.PackedConv(w, b, stride=st, padding=(KH // 2, KW // 2))
    out = ops.conv2d(pk, x)
    for _ in range(3):
        ops.conv2d(pk, x, out=out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.conv2d(pk, x, out=out)
    e.record()
    torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / reps
    fl = 2.0 * Cin * KH * KW * Cout * B * out.shape[2] * out.shape[3]
    print("%-36s %9.1f us  %7.2f TFLOP/s" % ("Cin%d Cout%d k%dx%d s%d B%d %dx%d" % (Cin, Cout, KH, KW, st, B, H, W), us,
                                              fl / us / 1e6), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--shapes", nargs="*", default=["big"])
    a = ap.parse_args()
    shapes = []
    for s in a.shapes:
        if s == "big":
            shapes += BIG
        elif s == "small":
            shapes += SMALL
        elif s == "all":
            shapes += BIG + SMALL
        else:
            shapes.append(tuple(int(v) for v in s.split(",")))
    for sh in shapes:
        run(sh, a.reps)
