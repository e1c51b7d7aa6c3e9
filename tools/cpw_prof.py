"""Where a step of the persistent correlation GEMM goes (csrc/corr_gemm_pw.hip, stamped instantiation): per workgroup the
cycles wave 0 (multiplying) and wave 4 (storing) spend working between barriers and waiting inside them."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accflow_amd import ops, _lib
lib = _lib.load()
f = lib.accflow_debug_cpw_prof
f.argtypes = [ctypes.c_void_p]
fm = torch.randn(7, 256, 60, 128, device="cuda")
idx1, idx2 = [2, 2, 1, 3, 3, 4, 4, 5, 5, 6, 6], [1, 0, 0, 2, 0, 3, 0, 4, 0, 5, 0]
with ops.conv_mode("f16x3"):
    packs = ops.corr_pack(fm)
    for _ in range(2):
        ops.corr_volume_disp_packed(packs, idx1, idx2)
    buf = torch.zeros(256 * 8, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    f(ctypes.c_void_p(buf.data_ptr()))
    ops.corr_volume_disp_packed(packs, idx1, idx2)
    torch.cuda.synchronize()
    f(ctypes.c_void_p(0))
t = buf.cpu().numpy().astype(np.float64).reshape(256, 8)
n = t[:, 2]
print("multiplying wave 0: work %.0f cycles per barrier interval, wait %.0f; %d intervals per workgroup, %.1f tiles" % (
    (t[:, 0] / n).mean(), (t[:, 1] / n).mean(), int(n.mean()), t[:, 3].mean()))
n4 = t[:, 6]
print("storing wave 4:     work %.0f cycles per barrier interval, wait %.0f" % ((t[:, 4] / n4).mean(), (t[:, 5] / n4).mean()))
