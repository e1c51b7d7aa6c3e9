"""Time of the encoders' 7x7 stride-2 stem (conv_stem7_kernel) at the benchmark's shapes: 7 and 6 images of 480 x 1024,
S16 + ReLU output (cnet / context) and raw output + InstanceNorm statistics (fnet).   python tools/stem_bench.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accflow_amd import ops

torch.manual_seed(0)
w = torch.randn(64, 3, 7, 7).cuda() * 0.1
b = torch.randn(64).cuda() * 0.1
pk = ops.PackedConv(w, b, stride=2, padding=3)
for B in (7, 6):
    x = torch.randn(B, 3, 480, 1024).cuda()
    out16 = ops.S16.empty(B, 64, 240, 512, x.device)

    def run16():
        ops.conv2d(pk, x, out16=out16, act=ops.ACT_RELU, fp32_out=False)

    def runstats():
        ops.conv2d(pk, x, want_stats=True)

    with ops.conv_mode("f16x3"):
        for name, fn in (("S16 + ReLU", run16), ("raw + stats", runstats)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print("stem B%d %-12s %.1f us per launch" % (B, name, 1e3 * e0.elapsed_time(e1) / 20))
