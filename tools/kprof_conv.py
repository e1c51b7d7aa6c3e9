import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from accflow_amd import ops, _lib
lib = _lib.load()
f = lib.accflow_debug_kprof
f.argtypes = [ctypes.c_void_p, ctypes.c_int]
occ = (ctypes.c_int * 16)()
n = lib.accflow_debug_occupancy(occ)
print("occupancy (workgroups/CU) direct<2,3> direct<2,2,f16> direct<1,3> | ldsKB/CU ldsKB/block regs/CU regs/block:", list(occ)[:n])
def run(shape, reps=1):
    Cin, Cout, KH, KW, st, B, H, W = shape
    x = torch.randn(B, Cin, H, W, device="cuda")
    w = torch.randn(Cout, Cin, KH, KW, device="cuda") * 0.05
    b = torch.randn(Cout, device="cuda")
    pk = ops.PackedConv(w, b, stride=st, padding=(KH // 2, KW // 2))
    out = ops.conv2d(pk, x)
    buf = (ctypes.c_ulonglong * (4096 * 16))()
    f(buf, 1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.conv2d(pk, x, out=out)
    e.record()
    torch.cuda.synchronize()
    f(buf, 1)
    import numpy as np
    arr = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16).astype(np.float64)
    live = arr[arr[:, 10] > 0]
    v = list(live.sum(0))
    v[12] = live[:, 12].min(); v[13] = live[:, 13].max()
    n = max(v[5], 1)
    names = ["A-load / gather issue", "wait frags (lgkmcnt0)", "mfma issue", "split+store patch (per chunk, /step)", "barrier (per chunk, /step)", "", "", "frag read issue"]
    tot = sum(v[:5]) + v[7]
    print("shape", shape, "%.1f us/launch" % (1e3 * s.elapsed_time(e) / reps))
    for i in (0, 7, 1, 2, 3, 4):
        print("  %-32s %8.1f cycles/step  %5.1f %%" % (names[i], v[i] / n, 100.0 * v[i] / tot))
    print("  waves %d, per-wave loop %.0f memtime ticks = %.1f us of s_memrealtime (100 MHz) -> %.3f GHz tick rate" % (v[10], v[6] / max(v[10], 1), v[8] / max(v[10], 1) / 100.0, v[6] / max(v[8], 1) * 0.1))
    print("  mean wave lifetime %.1f us (realtime); kernel span first-entry..last-exit %.1f us" % (v[11] / max(v[10], 1) / 100.0, (v[13] - v[12]) / 100.0))
    print("  prologue %.1f us; entry->epilogue stores issued %.1f us" % (v[14] / max(v[10], 1) / 100.0, v[15] / max(v[10], 1) / 100.0))
    print("  total per step %.1f ; steps/wave %.1f ; loop cycles per wave %.0f" % (tot / n, 0, v[6] / (n / (Cin // 16 * KH * KW))))
import sys as _s
shapes = [tuple(int(v) for v in a.split(',')) for a in _s.argv[1:]] or [(128, 256, 3, 3, 1, 11, 60, 128), (384, 256, 1, 5, 1, 11, 60, 128)]
for sh in shapes:
    run(sh)
