"""In-kernel stamps of the correlation GEMM (-DACCFLOW_KPROF build): mean workgroup time in the K loop, in the displaced
store phase (until its last store instruction is issued) and until the stores are acknowledged; s_memrealtime ticks of 10 ns."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from accflow_amd import ops, _lib
lib = _lib.load()
f = lib.accflow_debug_kprof
f.argtypes = [ctypes.c_void_p, ctypes.c_int]
for (H8, W8) in [(60, 128), (90, 160)]:
    fm = torch.randn(2, 256, H8, W8, device="cuda")
    packs = ops.corr_pack(fm)
    ops.corr_volume_disp_packed(packs, [1], [0])
    buf = (ctypes.c_ulonglong * (4096 * 16))()
    f(buf, 1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.corr_volume_disp_packed(packs, [1], [0])
    e.record()
    torch.cuda.synchronize()
    f(buf, 1)
    arr = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16).astype(np.float64)
    live = arr[arr[:, 10] > 0]
    print("%dx%d one pair (GEMM + pooling) %.1f us; %d sampled workgroups: K loop %.2f us, store phase issue %.2f us, "
          "lifetime %.2f us (mean)" % (H8, W8, 1e3 * s.elapsed_time(e), len(live), live[:, 8].mean() / 100, live[:, 9].mean() / 100,
                                      live[:, 11].mean() / 100))
