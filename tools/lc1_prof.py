"""In-kernel timeline of the fused lookup -> convc1 kernel (the stamped instantiation, accflow_debug_lc1_prof): per wave
s_memrealtime stamps (100 MHz) at entry, role set-up, after every workgroup barrier, and exit.  Prints where a workgroup's
lifetime goes per role.  The hook exists in a TOOLS build only (the product library has no mutable global):
    cp -a accflow_amd/lib /tmp/lib_lc1prof
    python -m accflow_amd.build --libdir=/tmp/lib_lc1prof --unit-define=corr_lookup_conv:ACCFLOW_LC1_PROF
    ACCFLOW_HIP_LIB=/tmp/lib_lc1prof/libaccflow_hip.so python tools/lc1_prof.py [sigma]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accflow_amd import ops, _lib  # noqa: E402


def main():
    sg = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
    lib = _lib.load()
    f = lib.accflow_debug_lc1_prof
    f.argtypes = [ctypes.c_void_p]
    B, C, h, w = 11, 256, 60, 128
    g = torch.Generator().manual_seed(5)
    fm = torch.randn(B + 1, C, h, w, generator=g).cuda()
    wgt = (torch.randn(256, 324, 1, 1, generator=g) * 0.05).cuda()
    bias = torch.randn(256, generator=g).cuda()
    with ops.conv_mode("f16x3"):
        packs = ops.corr_pack(fm)
        pyr = ops.corr_volume_disp_packed(packs, list(range(1, B + 1)), [0] * B)
        pkf = ops.PackedConv(ops.lookup_fused_weight(wgt), bias)
        o16 = ops.S16.empty(B, 256, h, w, fm.device, zero=True)
        grid = ops.coords_grid(B, h, w, fm.device)
        smooth = torch.nn.functional.interpolate(3.0 * torch.randn(B, 2, 4, 6, generator=g), size=(h, w), mode="bilinear",
                                                 align_corners=True).cuda()
        coords = (grid + smooth + sg * torch.randn(B, 2, h, w, generator=g).cuda()).contiguous()
        for _ in range(3):
            ops.corr_lookup_convc1(pyr, coords, pkf, out16=o16)
        nwg = B * (h * w // 64)
        buf = torch.zeros(nwg * 8 * 16, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        f(ctypes.c_void_p(buf.data_ptr()))
        ops.corr_lookup_convc1(pyr, coords, pkf, out16=o16)
        torch.cuda.synchronize()
        f(ctypes.c_void_p(0))
    t = buf.cpu().numpy().astype(np.float64).reshape(nwg, 8, 16) / 100.0       # us
    t0 = t[:, :, 0].min()
    print("kernel span (first entry .. last exit): %.1f us, %d workgroups" % (t[:, :, 12].max() - t0, nwg))
    ent = t[:, 0, 0] - t0
    print("workgroup entry times: first round (< 2 us) %d workgroups; median entry %.1f us; last entry %.1f us"
          % (int((ent < 2.0).sum()), float(np.median(ent)), float(ent.max())))
    life = t[:, :, 12].max(axis=1) - t[:, :, 0].min(axis=1)
    print("workgroup lifetime: mean %.1f us, p10 %.1f, p90 %.1f" % (life.mean(), np.percentile(life, 10), np.percentile(life, 90)))
    for name, wv in (("sampler level 0", 0), ("sampler level 3", 3), ("multiplier 0", 4)):
        x = t[:, wv]
        setup = (x[:, 13] - x[:, 0]).mean()
        first = (x[:, 1] - x[:, 0]).mean()
        steps = np.diff(x[:, 1:12], axis=1)              # barrier c -> barrier c+1, 10 intervals
        tail = (x[:, 12] - x[:, 11]).mean()
        print("%-16s set-up %.2f us | entry -> barrier 0 %.2f us | barrier-to-barrier mean %.2f us (per super-step: %s) | "
              "barrier 10 -> exit %.2f us" % (name, setup, first, steps.mean(), " ".join("%.2f" % v for v in steps.mean(axis=0)), tail))


if __name__ == "__main__":
    main()
