"""Helpers the estimators import (reference: networks/raft/utils/utils.py:66-126), HIP-backed."""
from .... import ops
from ..._packs import require_cuda


def coords_grid(batch, ht, wd, device):
    """(batch, 2, ht, wd): channel 0 = x, channel 1 = y (reference utils.py:83-87)."""
    return ops.coords_grid(batch, ht, wd, device)


def backwarp(image, flow, interp_mode="bilinear", padding_mode="zeros"):
    """Bilinear backward warp, zeros padding, align_corners (reference utils.py:96-124)."""
    if interp_mode != "bilinear" or padding_mode != "zeros":
        raise NotImplementedError("only bilinear / zeros is on the AccFlow inference path")
    require_cuda(image, flow)
    return ops.backwarp(image.float(), flow.float())
