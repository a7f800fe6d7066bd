"""Multi-head attention over SparseTensors (or dense (B, L, C) tensors) with the constructor surface and parameter names of
the reference's model/sparse_attention/modules.py:56-185, so its state dicts load unchanged:
`to_qkv` | (`to_q`, `to_kv`), `to_out`, `q_rms_norm.gamma` / `k_rms_norm.gamma`.

How a call runs here: the projections are the 16-bit MFMA GEMM (the input's own fp16 / bf16, else ops/precision.py; fp32 accumulate, fp32 bias), the channels are viewed as
[q|k|v][head][c] (or, with `use_old_attn_impl`, the older [head][q|k|v][c] of sparse/attention/modules.py:150-162), and the
token lists go to the varlen flash kernel directly or after the window / serialisation gather.  QK-RMSNorm is not a separate
pass: the per-head gains are handed to the kernel, which normalises q and k in its prologue.  RoPE is not built."""
from typing import *

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..basic import SparseTensor
from ..linear import linear_rows
from .full_attn import packed_varlen_attention
from .serialized_attn import SerializeMode, calc_serialization
from .windowed_attn import calc_window_partition

__all__ = ["SparseMultiHeadRMSNorm", "SparseMultiHeadAttention"]

Tokens = Union[SparseTensor, torch.Tensor]


class SparseMultiHeadRMSNorm(nn.Module):
    """x / |x| * gamma[head] * sqrt(dim) over the last axis, computed in fp32 (modules.py:56-69).  Holds the gains; inside
    SparseMultiHeadAttention the attention kernel applies them."""

    def __init__(self, dim: int, heads: int):
        super().__init__()
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.ones(heads, dim))

    def forward(self, x: Tokens) -> Tokens:
        f = x.feats if isinstance(x, SparseTensor) else x
        y = (F.normalize(f.float(), dim=-1) * self.gamma * self.scale).to(f.dtype)
        return x.replace(y) if isinstance(x, SparseTensor) else y


def _rows(x: Tokens) -> Tuple[torch.Tensor, List[int]]:
    """-> (token rows (T, C), tokens per batch element)."""
    if isinstance(x, SparseTensor):
        return x.feats, [s.stop - s.start for s in x.layout]
    return x.reshape(-1, x.shape[-1]), [x.shape[1]] * x.shape[0]


class SparseMultiHeadAttention(nn.Module):
    def __init__(self, channels: int, num_heads: int, ctx_channels: Optional[int] = None,
                 type: Literal["self", "cross"] = "self",
                 attn_mode: Literal["full", "serialized", "windowed"] = "full", window_size: Optional[int] = None,
                 shift_sequence: Optional[int] = None, shift_window: Optional[Tuple[int, int, int]] = None,
                 serialize_mode: Optional[SerializeMode] = None, qkv_bias: bool = True, use_rope: bool = False,
                 qk_rms_norm: bool = False, use_old_attn_impl: bool = False):
        super().__init__()
        if channels % num_heads:
            raise AssertionError("channels must be a multiple of num_heads")
        if type not in ("self", "cross"):
            raise AssertionError(f"Invalid attention type: {type}")
        if attn_mode not in ("full", "serialized", "windowed"):
            raise AssertionError(f"Invalid attention mode: {attn_mode}")
        if type == "cross" and attn_mode != "full":
            raise AssertionError("Cross-attention only supports full attention")
        if use_rope:
            raise NotImplementedError("RoPE is not built")
        self.channels, self.num_heads = channels, num_heads
        self.ctx_channels = channels if ctx_channels is None else ctx_channels
        self._type, self.attn_mode = type, attn_mode
        self.window_size, self.shift_sequence, self.shift_window = window_size, shift_sequence, shift_window
        self.serialize_mode = serialize_mode
        self.use_rope, self.qk_rms_norm, self.use_old_attn_impl = use_rope, qk_rms_norm, use_old_attn_impl
        if type == "self":
            self.to_qkv = nn.Linear(channels, 3 * channels, bias=qkv_bias)
        else:
            self.to_q = nn.Linear(channels, channels, bias=qkv_bias)
            self.to_kv = nn.Linear(self.ctx_channels, 2 * channels, bias=qkv_bias)
        if qk_rms_norm:
            self.q_rms_norm = SparseMultiHeadRMSNorm(channels // num_heads, num_heads)
            self.k_rms_norm = SparseMultiHeadRMSNorm(channels // num_heads, num_heads)
        self.to_out = nn.Linear(channels, channels)

    # ---- pieces ------------------------------------------------------------------------------------------------------
    def _split(self, rows: torch.Tensor, parts: int) -> List[torch.Tensor]:
        """(T, parts * channels) projection output -> `parts` views (T, H, c) in the configured channel layout."""
        T, H = rows.shape[0], self.num_heads
        if self.use_old_attn_impl:
            return list(rows.reshape(T, H, parts, -1).unbind(dim=2))
        return list(rows.reshape(T, parts, H, -1).unbind(dim=1))

    def _gains(self):
        if not self.qk_rms_norm:
            return None, None
        return self.q_rms_norm.gamma.detach().float().contiguous(), self.k_rms_norm.gamma.detach().float().contiguous()

    def _token_order(self, x: Tokens, lens: List[int]):
        """(gather index or None, scatter index or None, sequence lengths) of the configured attention mode."""
        if self.attn_mode == "full":
            return None, None, lens
        if not isinstance(x, SparseTensor):
            raise TypeError(f"{self.attn_mode} attention needs voxel coordinates: pass a SparseTensor")
        name = f"order_{self.attn_mode}_{self.window_size}_{self.shift_sequence}_{self.shift_window}_{self.serialize_mode}"
        hit = x.get_spatial_cache(name)
        if hit is None:
            if self.attn_mode == "windowed":
                fwd, bwd, seq, _ = calc_window_partition(x, self.window_size, self.shift_window if self.shift_window is not None else 0)
            else:
                fwd, bwd, seq, _ = calc_serialization(x, self.window_size, self.serialize_mode or SerializeMode.Z_ORDER,
                                                      self.shift_sequence or 0, self.shift_window or (0, 0, 0))
            hit = (fwd, bwd, seq)
            x.register_spatial_cache(name, hit)
        return hit

    # ---- call --------------------------------------------------------------------------------------------------------
    def forward(self, x: Tokens, context: Optional[Tokens] = None) -> Tokens:
        xr, q_lens = _rows(x)
        gq, gk = self._gains()
        if self._type == "self":
            q, k, v = self._split(linear_rows(self.to_qkv, xr).to(xr.dtype), 3)
            fwd, bwd, seq = self._token_order(x, q_lens)
            if fwd is not None:
                q, k, v = q[fwd], k[fwd], v[fwd]
            out = packed_varlen_attention(q, k, v, seq, seq, gq, gk)
            if bwd is not None:
                out = out[bwd]
        else:
            if context is None:
                raise ValueError("cross attention needs a context")
            cr, kv_lens = _rows(context)
            if len(kv_lens) != len(q_lens):
                raise AssertionError(f"Batch size mismatch, got {len(q_lens)} and {len(kv_lens)}")
            q = linear_rows(self.to_q, xr).to(xr.dtype).reshape(xr.shape[0], self.num_heads, -1)
            k, v = self._split(linear_rows(self.to_kv, cr).to(xr.dtype), 2)
            out = packed_varlen_attention(q, k, v, q_lens, kv_lens, gq, gk)
        y = linear_rows(self.to_out, out.reshape(out.shape[0], -1)).to(xr.dtype)
        return x.replace(y) if isinstance(x, SparseTensor) else y.reshape(*x.shape[:-1], -1)
