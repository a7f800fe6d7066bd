"""Per-sample normalisation layers over SparseTensor features (sparse/norm.py:12-41): every batch element's voxel list is
treated as one (1, C, L) signal.  Container-level plumbing on torch ops (no kernel of its own): the transformer blocks on the
path use the fused LayerNorm kernel instead; these exist for the reference's conv / flow-model side of `sparse/`."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .basic import SparseTensor

__all__ = ["SparseGroupNorm", "SparseLayerNorm"]


def _per_sample(input: SparseTensor, fn) -> SparseTensor:
    out = torch.zeros_like(input.feats)
    C = input.shape[1]
    for sl in input.layout:
        x = input.feats[sl].permute(1, 0).reshape(1, C, -1)
        out[sl] = fn(x).reshape(C, -1).permute(1, 0)
    return input.replace(out)


class SparseGroupNorm(nn.GroupNorm):
    def __init__(self, num_groups, num_channels, eps=1e-5, affine=True):
        super().__init__(num_groups, num_channels, eps, affine)

    def forward(self, input: SparseTensor) -> SparseTensor:
        return _per_sample(input, lambda x: F.group_norm(x, self.num_groups, self.weight, self.bias, self.eps))


class SparseLayerNorm(nn.LayerNorm):
    """NB upstream feeds the (1, C, L) view to nn.LayerNorm, i.e. normalises over the LAST axis of that view -- the L voxels of
    the sample -- with `normalized_shape` having to equal L; kept as is."""

    def __init__(self, normalized_shape, eps=1e-5, elementwise_affine=True):
        super().__init__(normalized_shape, eps, elementwise_affine)

    def forward(self, input: SparseTensor) -> SparseTensor:
        return _per_sample(input, lambda x: F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps))
