"""Resolution changes of a SparseTensor (sparse/spatial.py:13-110): average-pool downsample, its cached nearest-neighbour
inverse, and 2x subdivision.  Index plumbing on torch ops (unique / scatter_reduce / gather), on whatever device the
tensor lives; used by the reference's conv / flow-model side of `sparse/`, not by the transformer path."""
from typing import *

import torch
import torch.nn as nn

from .basic import SparseTensor

__all__ = ["SparseDownsample", "SparseUpsample", "SparseSubdivide"]


def _factor(factor, dim):
    f = tuple(factor) if isinstance(factor, (list, tuple)) else (factor,) * dim
    assert len(f) == dim, "Input coordinates must have the same dimension as the resampling factor."
    return f


class SparseDownsample(nn.Module):
    def __init__(self, factor: Union[int, Tuple[int, ...], List[int]]):
        super().__init__()
        self.factor = tuple(factor) if isinstance(factor, (list, tuple)) else factor

    def forward(self, input: SparseTensor) -> SparseTensor:
        dim = input.coords.shape[-1] - 1
        factor = _factor(self.factor, dim)
        c = input.coords.long().clone()
        c[:, 1:] = torch.div(c[:, 1:], torch.tensor(factor, device=c.device), rounding_mode="floor")
        extent = [int(c[:, i + 1].max()) + 1 for i in range(dim)]
        stride = [1]
        for e in extent[::-1]:
            stride.insert(0, stride[0] * e)                              # [batch stride, x stride, ..., 1]
        code = (c * torch.tensor(stride, device=c.device)).sum(dim=1)
        code, idx = code.unique(return_inverse=True)                     # ascending code = batch-major: batches stay contiguous
        C = input.feats.shape[1]
        pooled = torch.zeros(code.shape[0], C, device=input.feats.device, dtype=input.feats.dtype).scatter_reduce(
            0, idx.unsqueeze(1).expand(-1, C), input.feats, reduce="mean")        # include_self: the zero row counts, as upstream
        coords = torch.stack([code // stride[0]] + [(code // stride[i + 1]) % extent[i] for i in range(dim)], dim=-1)
        out = SparseTensor(pooled, coords.int(), input.shape, scale=tuple(s // f for s, f in zip(input._scale, factor)),
                           spatial_cache=input._spatial_cache)
        out.register_spatial_cache(f"upsample_{factor}_coords", input.coords)
        out.register_spatial_cache(f"upsample_{factor}_layout", input.layout)
        out.register_spatial_cache(f"upsample_{factor}_idx", idx)
        return out


class SparseUpsample(nn.Module):
    def __init__(self, factor: Union[int, Tuple[int, int, int], List[int]]):
        super().__init__()
        self.factor = tuple(factor) if isinstance(factor, (list, tuple)) else factor

    def forward(self, input: SparseTensor) -> SparseTensor:
        factor = _factor(self.factor, input.coords.shape[-1] - 1)
        coords, layout, idx = (input.get_spatial_cache(f"upsample_{factor}_{k}") for k in ("coords", "layout", "idx"))
        if coords is None or layout is None or idx is None:
            raise ValueError("Upsample cache not found. SparseUpsample must be paired with SparseDownsample.")
        return SparseTensor(input.feats[idx], coords, input.shape, layout, scale=tuple(s * f for s, f in zip(input._scale, factor)),
                            spatial_cache=input._spatial_cache)


class SparseSubdivide(nn.Module):
    def forward(self, input: SparseTensor) -> SparseTensor:
        dim = input.coords.shape[-1] - 1
        corners = torch.nonzero(torch.ones([2] * dim, device=input.device, dtype=torch.int))     # (2^dim, dim), lexicographic
        corners = torch.cat([torch.zeros_like(corners[:, :1]), corners], dim=-1).to(input.coords.dtype)
        base = input.coords.clone()
        base[:, 1:] *= 2
        coords = (base.unsqueeze(1) + corners.unsqueeze(0)).flatten(0, 1)
        feats = input.feats.unsqueeze(1).expand(input.feats.shape[0], corners.shape[0], *input.feats.shape[1:]).flatten(0, 1)
        # upstream: `out._scale = input._scale * 2` -- tuple repetition ((1,1,1) -> (1,1,1,1,1,1)), kept: it only keys the cache
        return SparseTensor(feats, coords, input.shape, scale=tuple(input._scale) * 2, spatial_cache=input._spatial_cache)
