"""The persistent correlation GEMM (csrc/corr_gemm_pw.hip: all pairs in one launch, the displaced store underneath the next
tile's matrix work) against the per-pair kernel it replaces (corr_disp_ring_kernel, ACCFLOW_CORR_GEMM=ring): the same
products in the same order and the same pooling order - all four pyramid levels must be BIT-IDENTICAL, at ragged sizes too
(a last query block of fewer than 128 pixels, odd heights, widths that are no multiple of 64, > 16 pairs = two launches;
C = 256 throughout: the persistent kernel is cut for 16 steps per tile, other channel counts run the per-pair kernel).
The library reads the switch once per process, so each side runs in a child process."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
from accflow_amd import ops
out = {}
for name, (F, C, h, w, pairs) in {"c3": (4, 256, 60, 128, [(1, 0), (2, 1), (2, 0), (3, 0), (3, 2)]),
                                   "ragged": (3, 256, 17, 22, [(1, 0), (2, 0)]),
                                   "odd": (3, 256, 9, 70, [(2, 1), (1, 0), (0, 2)]),
                                   "many": (5, 256, 16, 24, [(i %% 5, (i * 3 + 1) %% 5) for i in range(19)])}.items():
    g = torch.Generator().manual_seed(7)
    fm = torch.randn(F, C, h, w, generator=g).cuda()
    with ops.conv_mode("f16x3"):
        pk = ops.corr_pack(fm)
        pyr = ops.corr_volume_disp_packed(pk, [p[0] for p in pairs], [p[1] for p in pairs])
    for l, t in enumerate(pyr.to_rowmajor()):     # (row-major view: the padding lanes of the last query block drop out)
        out["%%s/%%d" %% (name, l)] = t.cpu()
torch.save(out, sys.argv[1])
"""


def _run(mode, path):
    env = dict(os.environ, ACCFLOW_CORR_GEMM=mode)
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT, path], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return torch.load(path)


def test_persistent_gemm_is_bit_identical_to_the_per_pair_kernel(tmp_path):
    a = _run("pw", str(tmp_path / "pw.pt"))
    b = _run("ring", str(tmp_path / "ring.pt"))
    assert sorted(a) == sorted(b) and len(a) == 16
    for k in a:
        assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k
    # and it is the volume: level 0 of the C3-shaped case against the fp32 matmul of the same features
    g = torch.Generator().manual_seed(7)
    fm = torch.randn(4, 256, 60, 128, generator=g)
    ref = torch.einsum("cp,cq->pq", fm[1].reshape(256, -1).double(), fm[0].reshape(256, -1).double()) / 16.0
    got = a["c3/0"][:7680].reshape(7680, 7680).double()        # pair (1, 0)
    assert float((got - ref).abs().max()) <= 2e-4 * float(ref.abs().max())
