"""Resolution changes of a SparseTensor -- the reference's sparse/spatial.py surface (SparseDownsample :13-58, SparseUpsample
:61-85, SparseSubdivide :87-110).  Index plumbing on torch ops, on whatever device the tensor lives; used by the conv /
flow-model side of `sparse/`, not by the transformer path.

Downsample: voxels are merged per (batch, coarse cell); the pooled feature is  sum / (count + 1)  -- upstream reduces with
`scatter_reduce(..., 'mean')` into a zero tensor whose zero row takes part in the mean, and that is kept.  The fine
coordinates, layout and fine->coarse index are left on the spatial cache so that Upsample can undo the merge exactly."""
from typing import *

import torch
import torch.nn as nn

from .basic import SparseTensor

__all__ = ["SparseDownsample", "SparseUpsample", "SparseSubdivide"]


def _per_axis(factor, ndim: int) -> Tuple[int, ...]:
    f = tuple(factor) if isinstance(factor, (list, tuple)) else (factor,) * ndim
    if len(f) != ndim:
        raise AssertionError("Input coordinates must have the same dimension as the resampling factor.")
    return f


def _cache_keys(factor: Tuple[int, ...]) -> Tuple[str, str, str]:
    return tuple(f"upsample_{factor}_{what}" for what in ("coords", "layout", "idx"))


class SparseDownsample(nn.Module):
    def __init__(self, factor: Union[int, Tuple[int, ...], List[int]]):
        super().__init__()
        self.factor = tuple(factor) if isinstance(factor, (list, tuple)) else factor

    def forward(self, input: SparseTensor) -> SparseTensor:
        ndim = input.coords.shape[-1] - 1
        factor = _per_axis(self.factor, ndim)
        dev = input.coords.device
        cell = input.coords.long().clone()
        cell[:, 1:] = torch.div(cell[:, 1:], torch.tensor(factor, device=dev), rounding_mode="floor")
        # mixed-radix key, batch index most significant: sorted unique keys keep the batches contiguous
        radix = (cell[:, 1:].amax(dim=0) + 1).tolist()
        key = cell[:, 0]
        for axis, r in enumerate(radix):
            key = key * r + cell[:, axis + 1]
        uniq, inverse = torch.unique(key, return_inverse=True)
        width = input.feats.shape[1]
        total = torch.zeros((uniq.shape[0], width), dtype=input.feats.dtype, device=input.feats.device).index_add_(0, inverse, input.feats)
        members = torch.bincount(inverse, minlength=uniq.shape[0]).to(total.dtype)
        pooled = total / (members + 1).unsqueeze(1)
        digits, rest = [], uniq
        for r in reversed(radix):
            digits.append(rest % r)
            rest = torch.div(rest, r, rounding_mode="floor")
        coarse = torch.stack([rest] + digits[::-1], dim=-1).int()
        out = SparseTensor(pooled, coarse, input.shape, scale=tuple(s // f for s, f in zip(input._scale, factor)),
                           spatial_cache=input._spatial_cache)
        for name, value in zip(_cache_keys(factor), (input.coords, input.layout, inverse)):
            out.register_spatial_cache(name, value)
        return out


class SparseUpsample(nn.Module):
    def __init__(self, factor: Union[int, Tuple[int, int, int], List[int]]):
        super().__init__()
        self.factor = tuple(factor) if isinstance(factor, (list, tuple)) else factor

    def forward(self, input: SparseTensor) -> SparseTensor:
        factor = _per_axis(self.factor, input.coords.shape[-1] - 1)
        fine_coords, fine_layout, to_coarse = (input.get_spatial_cache(k) for k in _cache_keys(factor))
        if fine_coords is None or fine_layout is None or to_coarse is None:
            raise ValueError("Upsample cache not found. SparseUpsample must be paired with SparseDownsample.")
        return SparseTensor(input.feats[to_coarse], fine_coords, input.shape, fine_layout,
                            scale=tuple(s * f for s, f in zip(input._scale, factor)), spatial_cache=input._spatial_cache)


class SparseSubdivide(nn.Module):
    """Every voxel becomes its 2^ndim children (coordinates doubled, children in lexicographic corner order), features repeated."""

    def forward(self, input: SparseTensor) -> SparseTensor:
        ndim = input.coords.shape[-1] - 1
        corner = torch.cartesian_prod(*[torch.arange(2, device=input.device)] * ndim).reshape(-1, ndim)
        child = input.coords.unsqueeze(1).repeat(1, corner.shape[0], 1)
        child[:, :, 1:] = child[:, :, 1:] * 2 + corner.to(child.dtype)
        feats = input.feats.unsqueeze(1).expand(-1, corner.shape[0], *input.feats.shape[1:])
        # upstream sets `_scale = input._scale * 2`: tuple repetition ((1,1,1) -> six ones); it only keys the cache, kept as is
        return SparseTensor(feats.flatten(0, 1), child.flatten(0, 1), input.shape, scale=tuple(input._scale) * 2,
                            spatial_cache=input._spatial_cache)
