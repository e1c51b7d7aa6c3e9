#!/usr/bin/env python3
"""Time of the displaced correlation GEMM (levels 0 + 1) + pooling for the C3 / C5 pair batches.
usage: [ACCFLOW_CORR_GEMM=regs] python tools/corr_gemm_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from accflow_amd import ops  # noqa: E402

for (F, H8, W8, pairs) in [(7, 60, 128, 11), (7, 90, 160, 11)]:
    fm = torch.randn(F, 256, H8, W8, device="cuda")
    packs = ops.corr_pack(fm)
    idx1 = [2, 2, 1, 3, 3, 4, 4, 5, 5, 6, 6][:pairs]
    idx2 = [1, 0, 0, 2, 0, 3, 0, 4, 0, 5, 0][:pairs]
    for _ in range(2):
        pyr = ops.corr_volume_disp_packed(packs, idx1, idx2)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    n = 5
    for _ in range(n):
        pyr = ops.corr_volume_disp_packed(packs, idx1, idx2)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    P = H8 * W8
    gb = pairs * (4.0 * P * P * (1 + 0.25 + 1 / 16 + 1 / 64)) / 1e9
    print("%dx%d, %d pairs: %.3f ms per batch (GEMM + pooling) = %.1f us per pair; %.2f TB/s of pyramid writes"
          % (H8, W8, pairs, ms, 1e3 * ms / pairs, gb / ms))
