import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accflow_amd import ops
g = torch.Generator().manual_seed(7)
for (Cin, Cout, KH, KW, B, H, W) in [(128, 256, 3, 3, 2, 20, 40), (128, 128, 1, 1, 1, 8, 32), (16, 128, 3, 3, 1, 4, 32), (32, 128, 3, 3, 1, 4, 32)]:
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, KH, KW, generator=g) * (1.0 / (Cin * KH * KW)) ** 0.5).cuda()
    pk = ops.PackedConv(w, None, padding=(KH // 2, KW // 2))
    want = ops.conv2d(pk, x)
    got = ops.conv2d(pk, ops.to_s16(x))
    d = (got - want).abs()
    print((Cin, Cout, KH, KW, B, H, W), "max diff %.3e" % float(d.max()), "frac differing %.4f" % float((d > 0).float().mean()))
    if float(d.max()) > 0:
        idx = (d > 0).nonzero()
        print("  first differing:", idx[:5].tolist(), " last:", idx[-3:].tolist())
        pb = (d > 0).float().sum(dim=(0, 1))
        print("  per-pixel count of differing channels (rows):", pb.sum(1).int().tolist()[:20])
        print("  per-column:", pb.sum(0).int().tolist()[:40])
