"""CorrBlock: all-pairs correlation pyramid + radius-r lookup on HIP kernels.

Same constructor / call contract as the reference's networks/raft/corr.py:7-55 (and gma/corr.py:8-58):
`CorrBlock(fmap1, fmap2, num_levels=4, radius=4)` builds `corr_pyramid` (list of (B*H*W, 1, Hl, Wl)
fp32 tensors) and `corr_fn(coords)` returns (B, num_levels*(2r+1)^2, H, W).
"""
from ... import ops
from .._packs import require_cuda


import os

# hot-path layout of the pyramid: "disp" (displacement-indexed, csrc/corr_disp.hip; default) or "row" (the reference's
# corr_pyramid layout, csrc/corr_lookup.hip).  Lookup results are the same.
LAYOUT = os.environ.get("ACCFLOW_CORR_LAYOUT", "disp")


class CorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        if num_levels != 4 or radius != 4:
            raise NotImplementedError("HIP CorrBlock is built for the reference's fixed 4 levels / radius 4 "
                                      "(raft.py:39-40, gma.py:21-22)")
        require_cuda(fmap1, fmap2)
        self.num_levels = num_levels
        self.radius = radius
        fmap1, fmap2 = fmap1.float().contiguous(), fmap2.float().contiguous()
        H8, W8 = fmap1.shape[-2:]
        # Measured on MI355X, B = 11 pairs at 60x128 (us per lookup launch): row-major 112-116 whatever the flow;
        # displaced 40 for coherent flow, 52 in the benchmark, ~115 for pure noise (a 4x8-tiled layout measured 135 and was removed).  The displaced volume
        # needs a split conv mode (its level 0 comes out of the matrix-core GEMM's displaced-store epilogue).
        def build():
            if LAYOUT == "disp" and ops.current_mode() != ops.CONV_F32 and ops.corr_disp_supported(H8, W8):
                return ops.corr_volume_disp(fmap1, fmap2)
            return ops.corr_volume(fmap1, fmap2)
        # (stand-alone use: the f16x3 volume is rebuilt in bf16x6 if a feature left the fp16 split's range; inside an
        # estimator forward the enclosing guarded region decides)
        self._pyr = ops.with_range_guard(build, fmap1.device)

    @classmethod
    def from_packs(cls, packs, idx1, idx2):
        """Pairs (query frame idx1[b], target frame idx2[b]) of per-frame operand packs (ops.corr_pack): what the
        estimators use when one call evaluates many pairs over few frames (AccFlow: 11 pairs, 7 frames)."""
        self = cls.__new__(cls)
        self.num_levels, self.radius = 4, 4
        self._pyr = ops.corr_volume_disp_packed(packs, idx1, idx2)
        return self

    @property
    def corr_pyramid(self):
        """list of (B*H*W, 1, Hl, Wl) tensors as in the reference (converted on demand from the hot-path layout)"""
        return self._pyr if isinstance(self._pyr, list) else self._pyr.to_rowmajor()

    def supports_s16(self):
        return isinstance(self._pyr, ops.DispPyramid)

    def lookup_s16(self, coords, out16, cache=None):
        """The lookup written pre-split for convc1's S16 pack (ops.corr_lookup_s16); cache: see ops.conv2d."""
        if cache is not None and cache[1] in cache[0]:
            return ops.corr_lookup_s16(self._pyr, coords, out16, cache=cache)
        require_cuda(coords)
        return ops.corr_lookup_s16(self._pyr, coords.float().contiguous(), out16, cache=cache)

    def lookup_convc1(self, coords, pk, out16, cache=None):
        """relu(convc1(lookup)) in one kernel (ops.corr_lookup_convc1): the taps never reach HBM; pk = the fused pack of
        convc1 (PackCache.conv(..., lookup_fused=True)); cache: see ops.conv2d."""
        if cache is not None and cache[1] in cache[0]:
            return ops.corr_lookup_convc1(self._pyr, coords, pk, out16=out16, cache=cache)
        require_cuda(coords)
        return ops.corr_lookup_convc1(self._pyr, coords.float().contiguous(), pk, out16=out16, act=ops.ACT_RELU, cache=cache)

    def __call__(self, coords, out=None):
        require_cuda(coords)
        return ops.corr_lookup(self._pyr, coords.float().contiguous(), out=out)

    @staticmethod
    def corr(fmap1, fmap2):
        """(B, H, W, 1, H, W) level-0 volume (reference corr.py:47-55)."""
        B, _, H, W = fmap1.shape
        lvl0 = ops.corr_volume(fmap1.float().contiguous(), fmap2.float().contiguous())[0]
        return lvl0.view(B, H, W, 1, H, W)
