"""In-kernel cycle stamps of the multi-source S16 kernel (-DACCFLOW_KPROF build, tools/bin/lib_kprof):
   ACCFLOW_HIP_LIB=tools/bin/lib_kprof/libaccflow_hip.so python tools/kprof_s16m.py lay:Cin,Cout,KH,KW,B,H,W ..."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from accflow_amd import _lib, ops  # noqa: E402

lib = _lib.load()


def run(lay, shape, reps=3):
    Cin, Cout, KH, KW, B, H, W = shape
    x = ops.to_s16(torch.randn(B, Cin, H, W, device="cuda"))
    w = torch.randn(Cout, Cin, KH, KW, device="cuda") * 0.05
    pk = ops.PackedMulti.from_cat(w, torch.randn(Cout, device="cuda"), [Cin], (KH // 2, KW // 2))
    out = torch.empty((B, Cout, H, W), device="cuda")
    f = getattr(lib, "accflow_debug_kprof_s16m_%d" % lay)
    f.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = (ctypes.c_ulonglong * (4096 * 16))()
    import time
    t_end = time.time() + 1.5          # reach the clock the chip holds under sustained load before stamping
    while time.time() < t_end:
        for _ in range(50):
            ops.conv2d_multi(pk, [x], out=out, lay=lay)
        torch.cuda.synchronize()
    f(buf, 1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.conv2d_multi(pk, [x], out=out, lay=lay)
    e.record()
    torch.cuda.synchronize()
    f(buf, 1)
    arr = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16).astype(np.float64)
    live = arr[arr[:, 10] > 0]
    n = len(live)
    v = live.mean(0)
    steps = v[7]
    print("lay %d shape %s: %.1f us/launch; %d workgroup records, %.0f steps per workgroup" % (lay, shape, 1e3 * s.elapsed_time(e) / reps, n, steps))
    names = ["A load + DMA issue", "B read issue", "wait frags (lgkmcnt 0)", "mfma issue", "vmcnt(0) at chunk end", "barrier at chunk end"]
    tot = v[8] + v[9] + v[11] + v[12]
    for i, nm in enumerate(names):
        print("  %-28s %9.1f cycles/step   %5.1f %% of the lifetime" % (nm, v[i] / steps, 100.0 * v[i] / tot))
    print("  prologue %.0f  loop %.0f (%.0f / step)  epilogue issue %.0f  store drain %.0f  cycles; lifetime %.0f" % (
        v[8], v[9], v[9] / steps, v[11], v[12], tot))
    if v[13] > 0:
        span = (live[:, 14] + live[:, 13]).max() - live[:, 14].min()
        print("  lifetime %.2f us of s_memrealtime -> clock %.3f GHz; first entry .. last exit %.1f us" % (
            v[13] / 100.0, tot / v[13] * 0.1, span / 100.0))


for a in sys.argv[1:]:
    lay, sh = a.split(":")
    run(int(lay), tuple(int(t) for t in sh.split(",")))
