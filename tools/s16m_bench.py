#!/usr/bin/env python3
"""Per-shape timing of the multi-source S16 kernel (csrc/conv_s16m_kernel.h): every wave layout on the workload's stride-1
shapes, the stride-2 convolutions as parity sources against the im2col kernel they replace, and AccPlus's 3- / 4-member
cats against copy + conv.   usage: python tools/s16m_bench.py [--reps 20] [--rounds 3]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from accflow_amd import ops  # noqa: E402

SHAPES = [
    # member channels, Cout, KH, KW, B, H, W
    ((64,), 64, 3, 3, 7, 240, 512),          # encoder layer1
    ((96,), 96, 3, 3, 7, 120, 256),          # encoder layer2
    ((128,), 128, 3, 3, 7, 60, 128),         # encoder layer3
    ((128,), 256, 1, 1, 7, 60, 128),         # encoder head
    ((128, 128), 256, 1, 5, 11, 60, 128),    # GRU zr
    ((128, 128), 128, 5, 1, 11, 60, 128),    # GRU q
    ((256,), 192, 3, 3, 11, 60, 128),        # convc2
    ((128,), 256, 3, 3, 11, 60, 128),        # flow head conv1
    ((256,), 126, 3, 3, 11, 60, 128),        # motion encoder conv
    ((128,), 64, 3, 3, 11, 60, 128),         # convf2
    ((352,), 256, 1, 1, 11, 60, 128),        # convc1
    ((16,), 128, 1, 7, 11, 60, 128),         # convf1 as 1x7
    ((256,), 18, 1, 1, 11, 60, 128),         # flow head conv2 taps
    ((128, 128, 1), 256, 3, 3, 1, 60, 128),  # AccPlus conv1[0]
    ((128, 128, 128, 128), 256, 3, 3, 1, 60, 128),  # AccPlus conv4[0]
    ((256,), 128, 3, 3, 1, 60, 128),
]
STRIDED = [
    # Cin, Cout, K, pad, B, H (input), W
    (64, 96, 3, 1, 7, 240, 512),
    (64, 96, 1, 0, 7, 240, 512),
    (96, 128, 3, 1, 7, 120, 256),
    (96, 128, 1, 0, 7, 120, 256),
]


def timeit(fn, reps, rounds):
    best = 1e9
    for _ in range(rounds):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * s.elapsed_time(e) / reps)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--only", default=None, help="comma-separated indices into SHAPES (and no strided shapes)")
    a = ap.parse_args()
    global SHAPES, STRIDED
    if a.only:
        SHAPES = [SHAPES[int(i)] for i in a.only.split(",")]
        STRIDED = []
    print("%-52s %9s %9s %9s %9s %9s   (us per launch; S16 in, fp32 out)" % ("shape", "auto", "lay0", "lay1", "lay2", "lay3"))
    for cs, Cout, KH, KW, B, H, W in SHAPES:
        Cin = sum(cs)
        xs = [ops.to_s16(torch.randn(B, c, H, W, device="cuda")) for c in cs]
        w = torch.randn(Cout, Cin, KH, KW, device="cuda") * 0.05
        pk = ops.PackedMulti.from_cat(w, torch.randn(Cout, device="cuda"), list(cs), (KH // 2, KW // 2))
        out = torch.empty((B, Cout, H, W), device="cuda")
        t = [timeit(lambda: ops.conv2d_multi(pk, xs, out=out, act=ops.ACT_RELU, lay=lay), a.reps, a.rounds)
             for lay in (None, 0, 1, 2, 3)]
        fl = 2.0 * Cin * KH * KW * Cout * B * H * W
        print("%-52s %9.1f %9.1f %9.1f %9.1f %9.1f   auto %6.1f best %6.1f TFLOP/s" % (
            "Cin%s Cout%d k%dx%d B%d %dx%d" % ("+".join(map(str, cs)), Cout, KH, KW, B, H, W), *t, fl / t[0] / 1e6,
            fl / min(t) / 1e6), flush=True)
    print("\n%-52s %9s %9s %9s %9s %9s %9s" % ("stride-2 shape", "im2col", "auto", "lay0", "lay1", "lay2", "lay3"))
    for Cin, Cout, K, p, B, H, W in STRIDED:
        x = torch.randn(B, Cin, H, W, device="cuda")
        w = torch.randn(Cout, Cin, K, K, device="cuda") * 0.05
        b = torch.randn(Cout, device="cuda")
        pk0 = ops.PackedConv(w, b, stride=2, padding=p)
        t0 = timeit(lambda: ops.conv2d(pk0, x), a.reps, a.rounds)
        pk = ops.PackedMulti.from_strided(w, b, p)
        x16 = ops.to_s16(x)
        OH, OW = pk0.out_size(H, W)
        out = torch.empty((B, Cout, OH, OW), device="cuda")
        t = [timeit(lambda: ops.conv2d_multi(pk, [x16] * len(pk.C), out=out, out_hw=(OH, OW), lay=lay), a.reps, a.rounds)
             for lay in (None, 0, 1, 2, 3)]
        fl = 2.0 * Cin * K * K * Cout * B * OH * OW
        print("%-52s %9.1f %9.1f %9.1f %9.1f %9.1f %9.1f   im2col %6.1f auto %6.1f TFLOP/s" % (
            "Cin%d Cout%d k%dx%d s2 B%d %dx%d" % (Cin, Cout, K, K, B, OH, OW), t0, *t, fl / t0 / 1e6, fl / t[0] / 1e6), flush=True)


if __name__ == "__main__":
    main()
