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


# ---- API helpers of the reference's utils module that are NOT on the inference hot path (test_cvo.py never imports them;
# the CorrBlock lookup, the warps and the upsampling run on their own HIP kernels).  They are kept for users of the
# reference's module surface (demo / Sintel scripts pad 436 x 1024 frames with InputPadder) and are plain tensor plumbing.
class InputPadder:
    """Pads images such that dimensions are divisible by 8 (reference utils.py:7-28): replicate padding, centred for
    mode 'sintel', bottom / centred-in-x otherwise; unpad() crops a result back."""

    def __init__(self, dims, mode="sintel"):
        self.ht, self.wd = dims[-2:]
        pad_ht = (((self.ht // 8) + 1) * 8 - self.ht) % 8
        pad_wd = (((self.wd // 8) + 1) * 8 - self.wd) % 8
        if mode == "sintel":
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]
        else:
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, 0, pad_ht]

    def pad(self, *inputs):
        import torch.nn.functional as F
        return [F.pad(x, self._pad, mode="replicate") for x in inputs]

    def unpad(self, x):
        ht, wd = x.shape[-2:]
        c = [self._pad[2], ht - self._pad[3], self._pad[0], wd - self._pad[1]]
        return x[..., c[0]:c[1], c[2]:c[3]]


def bilinear_sampler(img, coords, mode="bilinear", mask=False):
    """img (N, C, H, W) sampled at PIXEL coordinates coords (N, H', W', 2) = (x, y), zeros outside, align_corners
    (reference utils.py:66-80).  When the sample grid has the image's size this is the backwarp kernel (flow = coords -
    pixel grid); other grid shapes go through torch.nn.functional.grid_sample like the reference."""
    if mode != "bilinear":
        raise NotImplementedError("bilinear_sampler: bilinear only")
    import torch
    H, W = img.shape[-2:]
    if img.is_cuda and tuple(coords.shape[1:3]) == (H, W):
        grid = ops.coords_grid(img.shape[0], H, W, img.device)
        out = ops.backwarp(img.float().contiguous(), (coords.permute(0, 3, 1, 2).float() - grid).contiguous())
    else:
        import torch.nn.functional as F
        xg, yg = coords.split([1, 1], dim=-1)
        out = F.grid_sample(img, torch.cat([2 * xg / (W - 1) - 1, 2 * yg / (H - 1) - 1], dim=-1), align_corners=True)
    if mask:
        xg, yg = coords.split([1, 1], dim=-1)
        xn, yn = 2 * xg / (W - 1) - 1, 2 * yg / (H - 1) - 1
        return out, ((xn > -1) & (yn > -1) & (xn < 1) & (yn < 1)).float()
    return out


def upflow8(flow, mode="bilinear"):
    """8 x bilinear (align_corners) upsampling of a flow field, values scaled by 8 (reference utils.py:90-93; the
    estimators use the learned convex upsampling instead, raft.py:81-92)."""
    import torch.nn.functional as F
    return 8 * F.interpolate(flow, size=(8 * flow.shape[2], 8 * flow.shape[3]), mode=mode, align_corners=True)
