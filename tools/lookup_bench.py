#!/usr/bin/env python3
"""Micro-benchmark of the CorrBlock lookup at the C3 working size (B pairs x 60x128 query pixels)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from accflow_amd import ops  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=11)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--h8", type=int, default=60)
    ap.add_argument("--w8", type=int, default=128)
    ap.add_argument("--flow", type=float, default=2.0, help="std of the random flow added to the grid (1/8-res px)")
    ap.add_argument("--layout", default="row", choices=["row", "disp"])
    ap.add_argument("--smooth", type=float, default=0.0,
                    help="std of a smooth (bilinearly upsampled 4x8 grid) flow component, 1/8-res px")
    ap.add_argument("--zoom", type=float, default=0.0, help="flow = zoom * (p - centre): a smooth flow with this gradient (px / px)")
    a = ap.parse_args()
    B, h, w = a.pairs, a.h8, a.w8
    g = torch.Generator(device="cuda").manual_seed(0)
    f1 = torch.randn(B, 256, h, w, device="cuda", generator=g)
    f2 = torch.randn(B, 256, h, w, device="cuda", generator=g)
    build = {"row": ops.corr_volume, "disp": ops.corr_volume_disp}[a.layout]
    pyr = build(f1, f2)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    pyr = build(f1, f2)
    e.record()
    torch.cuda.synchronize()
    print("volume[%s] B=%d: %.2f ms (level 0 + 3 pooled levels)" % (a.layout, B, s.elapsed_time(e)))
    coords = ops.coords_grid(B, h, w, "cuda") + a.flow * torch.randn(B, 2, h, w, device="cuda", generator=g)
    if a.smooth > 0:
        coords = coords + torch.nn.functional.interpolate(a.smooth * torch.randn(B, 2, 4, 8, device="cuda", generator=g),
                                                          size=(h, w), mode="bilinear", align_corners=True)
    if a.zoom != 0.0:
        grid = ops.coords_grid(B, h, w, "cuda")
        centre = torch.tensor([(w - 1) / 2.0, (h - 1) / 2.0], device="cuda").view(1, 2, 1, 1)
        coords = coords + a.zoom * (grid - centre)
    out = ops.corr_lookup(pyr, coords)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.reps):
        ops.corr_lookup(pyr, coords, out=out)
    e.record()
    torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / a.reps
    by = ops.LOOKUP_BYTES_PER_PX * B * h * w
    print("lookup[%s, noise %.2f, smooth %.2f, zoom %.2f] B=%d %dx%d: %.1f us/launch, %.1f GB/s algorithmic (%d B/launch), %.1f%% of 8 TB/s" % (
        a.layout, a.flow, a.smooth, a.zoom, B, h, w, us, by / us / 1e3, by, 100 * by / us / 1e3 / 8000))
