"""CorrBlock: all-pairs correlation pyramid + radius-r lookup on HIP kernels.

Same constructor / call contract as the reference's networks/raft/corr.py:7-55 (and gma/corr.py:8-58):
`CorrBlock(fmap1, fmap2, num_levels=4, radius=4)` builds `corr_pyramid` (list of (B*H*W, 1, Hl, Wl)
fp32 tensors) and `corr_fn(coords)` returns (B, num_levels*(2r+1)^2, H, W).
"""
from ... import ops
from .._packs import require_cuda


import os

TILED = os.environ.get("ACCFLOW_CORR_TILED", "0") == "1"


class CorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        if num_levels != 4 or radius != 4:
            raise NotImplementedError("HIP CorrBlock is built for the reference's fixed 4 levels / radius 4 "
                                      "(raft.py:39-40, gma.py:21-22)")
        require_cuda(fmap1, fmap2)
        self.num_levels = num_levels
        self.radius = radius
        # Row-major planes (the reference's corr_pyramid layout).  A 4x8-tiled layout exists too
        # (ops.corr_volume_tiled, csrc/corr_tiled.hip: ~25 % less HBM traffic per lookup) but its lookup measured
        # 135 us vs 110 us per launch on MI355X - L1 thrash on the re-visited sectors - so it is not the default.
        if TILED:
            self._pyr = ops.corr_volume_tiled(fmap1.float().contiguous(), fmap2.float().contiguous())
            self.corr_pyramid = None
        else:
            self._pyr = self.corr_pyramid = ops.corr_volume(fmap1.float().contiguous(), fmap2.float().contiguous())

    def __call__(self, coords, out=None):
        require_cuda(coords)
        return ops.corr_lookup(self._pyr, coords.float().contiguous(), out=out)

    @staticmethod
    def corr(fmap1, fmap2):
        """(B, H, W, 1, H, W) level-0 volume (reference corr.py:47-55)."""
        B, _, H, W = fmap1.shape
        lvl0 = ops.corr_volume(fmap1.float().contiguous(), fmap2.float().contiguous())[0]
        return lvl0.view(B, H, W, 1, H, W)
