#!/usr/bin/env python3
"""Characterise the CorrBlock lookup (displaced layout) instead of quoting one point: time per B = 11 launch at 60x128
over the flow field's incoherence - i.i.d. noise sigma (1/8-res px) on top of a smooth component - for the fp32 output
(324 channels) and the S16 output (4 x 88 pre-split channels) the update block consumes.  Algorithmic bytes = 2 904 B per
query pixel; peak 8 TB/s.   usage: python tools/lookup_sweep.py [--fused] > profiles/rNN_lookup_sweep.txt
--fused: a third column - the kernel the product path runs, CorrBlock lookup FUSED with convc1
(accflow_corr_lookup_convc1_s16, csrc/corr_lookup_conv.hip): its algorithmic bytes are 1 608 B read + 1 024 B written per query
pixel (the 324 taps never reach HBM); the fraction printed for it is (1 608 + 1 024) B x pixels / time / 8 TB/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from accflow_amd import ops  # noqa: E402

B, h, w, reps = 11, 60, 128, 20
g = torch.Generator(device="cuda").manual_seed(0)
f1 = torch.randn(B, 256, h, w, device="cuda", generator=g)
f2 = torch.randn(B, 256, h, w, device="cuda", generator=g)
pyr = ops.corr_volume_disp(f1, f2)
out = torch.empty((B, 324, h, w), device="cuda")
out16 = ops.S16.empty(B, ops.LOOKUP_S16_CHANNELS, h, w, f1.device)
by = ops.LOOKUP_BYTES_PER_PX * B * h * w
FUSED = "--fused" in sys.argv
if FUSED:
    wgt = (torch.randn(256, 324, 1, 1, device="cuda", generator=g) * 0.05)
    pkf = ops.PackedConv(ops.lookup_fused_weight(wgt), torch.randn(256, device="cuda", generator=g))
    o16f = ops.S16.empty(B, 256, h, w, f1.device)
    byf = (1608 + 1024) * B * h * w


def timed(fn):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * s.elapsed_time(e) / reps)
    return best


print("corr_lookup_disp_kernel, B = %d pairs x %dx%d query pixels, %d B algorithmic per launch; us per launch (fraction of 8 TB/s)" % (B, h, w, by))
print("%-34s %22s %22s%s" % ("flow = grid + smooth + noise", "fp32 out (324 ch)", "S16 out (4 x 88 ch)",
                            "   fused lookup -> convc1 (%d B algorithmic)" % byf if FUSED else ""))
for smooth in (0.0, 4.0):
    for sigma in (0.0, 0.1, 0.5, 1.0, 2.0, 4.0):
        coords = ops.coords_grid(B, h, w, "cuda") + sigma * torch.randn(B, 2, h, w, device="cuda", generator=g)
        if smooth > 0:
            coords = coords + torch.nn.functional.interpolate(smooth * torch.randn(B, 2, 4, 8, device="cuda", generator=g),
                                                              size=(h, w), mode="bilinear", align_corners=True)
        coords = coords.contiguous()
        a = timed(lambda: ops.corr_lookup(pyr, coords, out=out))
        b = timed(lambda: ops.corr_lookup_s16(pyr, coords, out16))
        line = "smooth sigma %.1f px, noise sigma %.1f px   %8.1f us  (%.3f)   %8.1f us  (%.3f)" % (
            smooth, sigma, a, by / a / 1e3 / 8000, b, by / b / 1e3 / 8000)
        if FUSED:
            c = timed(lambda: ops.corr_lookup_convc1(pyr, coords, pkf, out16=o16f))
            line += "   %8.1f us  (%.3f)" % (c, byf / c / 1e3 / 8000)
        print(line, flush=True)
print("(the driver's bench: flows of a random-init estimator after 1..12 iterations - between the sigma 0.5 and 1.0 rows;"
      " a trained estimator's flows are piecewise smooth: the sigma <= 0.1 rows)")
