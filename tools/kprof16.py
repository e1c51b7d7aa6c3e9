"""In-kernel clock and cycles per step of the 16x16x32 direct conv kernel (library built with -DACCFLOW_KPROF)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from accflow_amd import ops, _lib
lib = _lib.load()
f16 = lib.accflow_debug_kprof16; f16.argtypes = [ctypes.c_void_p, ctypes.c_int]
f32 = lib.accflow_debug_kprof; f32.argtypes = [ctypes.c_void_p, ctypes.c_int]
def run(shape, reps=20):
    Cin, Cout, KH, KW, st, B, H, W = shape
    x = torch.randn(B, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, KH, KW, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
    pk = ops.PackedConv(w, b, stride=st, padding=(KH // 2, KW // 2))
    out = ops.conv2d(pk, x)
    for _ in range(reps): ops.conv2d(pk, x, out=out)
    buf = (ctypes.c_ulonglong * (4096 * 16))()
    f16(buf, 1); f32(buf, 1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): ops.conv2d(pk, x, out=out)
    e.record(); torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / reps
    f16(buf, 0)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16).astype(np.float64); a = a[a[:, 10] > 0]
    if len(a):
        print("  16x16x32 kernel: %.1f us/launch, loop clock %.3f GHz, %.0f cycles per 32-deep step" % (us, a[:, 0].sum() / a[:, 1].sum() * 0.1, (a[:, 0] / a[:, 2]).mean()))
    f32(buf, 0)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16).astype(np.float64); a = a[a[:, 10] > 0]
    if len(a):
        print("  32x32x16 kernel: %.1f us/launch, loop clock %.3f GHz, %.0f cycles per 16-deep step" % (us, a[:, 6].sum() / a[:, 8].sum() * 0.1, (a[:, 6] / np.maximum(a[:, 5], 1)).mean()))
for sh in [(128, 256, 3, 3, 1, 11, 60, 128), (384, 256, 1, 5, 1, 11, 60, 128), (384, 128, 5, 1, 1, 11, 60, 128)]:
    print(sh); run(sh)
