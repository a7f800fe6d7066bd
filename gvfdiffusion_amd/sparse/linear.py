"""SparseLinear (sparse/linear.py:10-15): nn.Linear over the feature rows of a SparseTensor, on the 16-bit MFMA GEMM."""
import torch
import torch.nn as nn

from .basic import SparseTensor
from ..ops import dit_ops, precision

__all__ = ["SparseLinear", "linear_rows"]


def _cached_weight(lin: nn.Linear, lp):
    """16-bit copy of the weight (K zero-padded to a multiple of 64) + the fp32 bias, converted once per parameter version and operand type
    (kept on the module; round 2 re-cast and re-padded them on every call)."""
    w = lin.weight
    ver = (w._version, w.data_ptr(), w.device, None if lin.bias is None else (lin.bias._version, lin.bias.data_ptr()))
    cache = lin.__dict__.setdefault("_gvf_wcache", {})
    hit = cache.get(lp)
    if hit is None or hit[0] != ver:
        wb = dit_ops.cast_pad(w.detach().float().contiguous(), dit_ops.pad64(w.shape[1]), dtype=lp)
        bias = None if lin.bias is None else lin.bias.detach().float().contiguous()
        hit = cache[lp] = (ver, wb, bias)
    return hit[1], hit[2]


def linear_rows(lin: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """x (T, K) -> fp32 (T, N): 16-bit operands (x's own type if it is fp16 / bf16, else ops/precision.py's rule; K zero-padded to a
    multiple of 64), fp32 accumulate, bias added in fp32."""
    lp = precision.resolve(None, (x,))
    K = x.shape[1]
    xb = dit_ops.cast_pad(x.float().contiguous(), dit_ops.pad64(K), dtype=lp)
    wb, bias = _cached_weight(lin, lp)
    out = torch.empty((x.shape[0], lin.out_features), dtype=torch.float32, device=x.device)
    return dit_ops.gemm(xb, wb, bias, out, dit_ops.EPI_STORE_F32)


class SparseLinear(nn.Linear):
    def forward(self, input: SparseTensor) -> SparseTensor:
        return input.replace(linear_rows(self, input.feats).to(input.dtype))
