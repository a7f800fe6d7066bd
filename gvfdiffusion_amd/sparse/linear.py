"""SparseLinear (sparse/linear.py:10-15): nn.Linear over the feature rows of a SparseTensor, on the bf16 MFMA GEMM."""
import torch
import torch.nn as nn

from .basic import SparseTensor
from ..ops import dit_ops

__all__ = ["SparseLinear", "linear_rows"]


def linear_rows(lin: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """x (T, K) float -> fp32 (T, N): bf16 operands (K zero-padded to a multiple of 64), fp32 accumulate, bias added in fp32."""
    K = x.shape[1]
    xb = dit_ops.cast_pad_bf16(x.float().contiguous(), dit_ops.pad64(K))
    wb = dit_ops.cast_pad_bf16(lin.weight.detach().float().contiguous(), dit_ops.pad64(K))
    out = torch.empty((x.shape[0], lin.out_features), dtype=torch.float32, device=x.device)
    bias = None if lin.bias is None else lin.bias.detach().float().contiguous()
    return dit_ops.gemm_bf16(xb, wb, bias, out, dit_ops.EPI_STORE_F32)


class SparseLinear(nn.Linear):
    def forward(self, input: SparseTensor) -> SparseTensor:
        return input.replace(linear_rows(self, input.feats).to(input.dtype))
