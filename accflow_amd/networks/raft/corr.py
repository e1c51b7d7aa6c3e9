"""CorrBlock: all-pairs correlation pyramid + radius-r lookup on HIP kernels.

Same constructor / call contract as the reference's networks/raft/corr.py:7-55 (and gma/corr.py:8-58):
`CorrBlock(fmap1, fmap2, num_levels=4, radius=4)` builds `corr_pyramid` (list of (B*H*W, 1, Hl, Wl)
fp32 tensors) and `corr_fn(coords)` returns (B, num_levels*(2r+1)^2, H, W).
"""
import torch

from ... import ops
from .._packs import require_cuda


class CorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        if num_levels != 4 or radius != 4:
            raise NotImplementedError("HIP CorrBlock is built for the reference's fixed 4 levels / radius 4 "
                                      "(raft.py:39-40, gma.py:21-22)")
        require_cuda(fmap1, fmap2)
        self.num_levels = num_levels
        self.radius = radius
        self.corr_pyramid = ops.corr_volume(fmap1.float().contiguous(), fmap2.float().contiguous())

    def __call__(self, coords, out=None):
        require_cuda(coords)
        return ops.corr_lookup(self.corr_pyramid, coords.float().contiguous(), out=out)

    @staticmethod
    def corr(fmap1, fmap2):
        """(B, H, W, 1, H, W) level-0 volume (reference corr.py:47-55)."""
        B, _, H, W = fmap1.shape
        lvl0 = ops.corr_volume(fmap1.float().contiguous(), fmap2.float().contiguous())[0]
        return lvl0.view(B, H, W, 1, H, W)
