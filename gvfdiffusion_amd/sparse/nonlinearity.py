"""Activations over SparseTensor features (the reference's sparse/nonlinearity.py surface: SparseReLU, SparseSiLU, SparseGELU,
SparseActivation).  Each class is its torch.nn activation applied to `feats`, re-wrapped with the input's coords / layout /
cache.  Stand-alone use only: inside the transformer blocks the GELU is the epilogue of the first MLP GEMM (csrc/gemm.hip)."""
import torch.nn as nn

from .basic import SparseTensor

__all__ = ["SparseReLU", "SparseSiLU", "SparseGELU", "SparseActivation"]


def _on_feats(base: type, name: str) -> type:
    """Subclass of a dense activation whose forward maps SparseTensor -> SparseTensor (constructor arguments unchanged)."""
    def forward(self, x: SparseTensor) -> SparseTensor:
        return x.replace(base.forward(self, x.feats))
    return type(name, (base,), {"forward": forward, "__doc__": f"{base.__name__} over the features of a SparseTensor.", "__module__": __name__})


SparseReLU = _on_feats(nn.ReLU, "SparseReLU")
SparseSiLU = _on_feats(nn.SiLU, "SparseSiLU")
SparseGELU = _on_feats(nn.GELU, "SparseGELU")


class SparseActivation(nn.Module):
    """Any dense activation module, applied to the features."""

    def __init__(self, activation: nn.Module):
        super().__init__()
        self.activation = activation

    def forward(self, x: SparseTensor) -> SparseTensor:
        return x.replace(self.activation(x.feats))
