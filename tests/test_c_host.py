"""The C-ABI from a plain C host (tests/c_host/host_corr.c): compiled with gcc against include/accflow_hip.h and
linked against libaccflow_hip.so + the ROCm HIP runtime - no Python, torch or device code on the host side.
CPU: it compiles and links (every symbol it uses is exported).  GPU: it runs the CorrBlock path in both layouts and the
results match the oracle (raft/corr.py:8-55 restated in oracle/accflow_oracle.py)."""
import os
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_host", "host_corr.c")


def _build(tmp_path):
    from accflow_amd import build as _b
    lib = _b.build(force=False, verbose=False)
    exe = str(tmp_path / "host_corr")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(rocm, "include"), SRC, "-o", exe,
           "-L" + os.path.dirname(lib), "-laccflow_hip", "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath," + os.path.join(rocm, "lib")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_host_compiles_and_links(tmp_path):
    exe = _build(tmp_path)
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True)   # usage message, no GPU call
    assert r.returncode == 1 and "usage" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 256, 16, 32), (1, 64, 12, 20)])
def test_c_host_corr_path_vs_oracle(tmp_path, shape):
    from oracle import accflow_oracle as O
    exe = _build(tmp_path)
    B, C, H8, W8 = shape
    g = torch.Generator().manual_seed(5)
    f1 = torch.randn(B, C, H8, W8, generator=g)
    f2 = torch.randn(B, C, H8, W8, generator=g)
    ys, xs = torch.meshgrid(torch.arange(H8, dtype=torch.float32), torch.arange(W8, dtype=torch.float32), indexing="ij")
    coords = torch.stack([xs, ys])[None].repeat(B, 1, 1, 1) + 3.0 * torch.randn(B, 2, H8, W8, generator=g)
    coords[0, :, 0, 0] = torch.tensor([-7.5, 2.0])          # out of the image
    coords[0, :, 1, 1] = torch.tensor([3.0, 4.0])           # integer position
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(np.array([B, C, H8, W8], dtype=np.int32).tobytes())
        for t in (f1, f2, coords):
            f.write(t.contiguous().numpy().astype(np.float32).tobytes())
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout, r.stderr)
    out = np.fromfile(tmp_path / "out.bin", dtype=np.float32).reshape(2, B, 324, H8, W8)
    pyr = O.corr_pyramid(f1, f2)
    ref = O.corr_lookup(pyr, coords).numpy()
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(out[0] - ref).max() <= 2e-5 * scale, ("reference layout", np.abs(out[0] - ref).max())
    assert np.abs(out[1] - ref).max() <= 5e-5 * scale, ("displaced layout", np.abs(out[1] - ref).max())
    print(r.stdout.strip())
