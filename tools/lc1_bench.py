"""Same-box timing of the fused lookup -> convc1 kernel against the two launches it replaces (S16 lookup, then convc1 on
the direct kernel), at the benchmark's shape (B = 11, 60 x 128) with the coordinates of a smooth flow + sigma px noise.
    python tools/lc1_bench.py [sigma ...]
"""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accflow_amd import ops  # noqa: E402


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n


def main():
    sigmas = [float(x) for x in sys.argv[1:]] or [0.0, 0.25, 1.0]
    B, C, h, w = 11, 256, 60, 128
    g = torch.Generator().manual_seed(5)
    f = torch.randn(B + 1, C, h, w, generator=g).cuda()
    wgt = (torch.randn(256, 324, 1, 1, generator=g) * 0.05).cuda()
    bias = torch.randn(256, generator=g).cuda()
    with ops.conv_mode("f16x3"):
        packs = ops.corr_pack(f)
        pyr = ops.corr_volume_disp_packed(packs, list(range(1, B + 1)), [0] * B)
        pkf = ops.PackedConv(ops.lookup_fused_weight(wgt), bias)
        w88 = torch.zeros(256, 4, 88, device="cuda")
        w88[:, :, :81] = wgt.reshape(256, 4, 9, 9).transpose(2, 3).reshape(256, 4, 81)
        pk2 = ops.PackedConv(w88.reshape(256, 352, 1, 1), bias)
        l16 = ops.S16.empty(B, ops.LOOKUP_S16_CHANNELS, h, w, f.device, zero=True)
        o16a = ops.S16.empty(B, 256, h, w, f.device, zero=True)
        o16b = ops.S16.empty(B, 256, h, w, f.device, zero=True)
        grid = ops.coords_grid(B, h, w, f.device)
        smooth = torch.nn.functional.interpolate(3.0 * torch.randn(B, 2, 4, 6, generator=g), size=(h, w), mode="bilinear",
                                                 align_corners=True).cuda()
        px = B * h * w
        for sg in sigmas:
            coords = (grid + smooth + sg * torch.randn(B, 2, h, w, generator=g).cuda()).contiguous()
            t_l = timeit(lambda: ops.corr_lookup_s16(pyr, coords, l16))
            t_c = timeit(lambda: ops.conv2d(pk2, l16, out16=o16b, act=ops.ACT_RELU, fp32_out=False))
            t_f = timeit(lambda: ops.corr_lookup_convc1(pyr, coords, pkf, out16=o16a))
            d = float((o16a.to_float() - o16b.to_float()).abs().max())
            print("sigma %.2f px: lookup %.1f us (%.2f of 8 TB/s) + convc1 %.1f us (%.0f TFLOP/s) = %.1f us | fused %.1f us "
                  "(%.0f TFLOP/s conv-equivalent, %.2f TB/s of the lookup's reads + the output) | max diff %.2e"
                  % (sg, t_l, 2904.0 * px / (t_l * 1e-6) / 8e12, t_c, 2.0 * 324 * 256 * px / (t_c * 1e-6) / 1e12, t_l + t_c,
                     t_f, 2.0 * 324 * 256 * px / (t_f * 1e-6) / 1e12, (1608.0 + 1024.0) * px / (t_f * 1e-6) / 1e12, d), flush=True)


if __name__ == "__main__":
    main()
