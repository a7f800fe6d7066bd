"""SparseMultiHeadRMSNorm / SparseMultiHeadAttention with the reference's constructor and parameter names
(model/sparse_attention/modules.py:56-185; the stray debug prints at :152,154 are not reproduced).
Projections run on the bf16 MFMA GEMM, attention on the varlen flash kernel; RoPE is not built."""
from typing import *

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..basic import SparseTensor
from .full_attn import sparse_scaled_dot_product_attention, packed_varlen_attention
from .serialized_attn import SerializeMode, sparse_serialized_scaled_dot_product_self_attention
from .windowed_attn import sparse_windowed_scaled_dot_product_self_attention
from ...ops import dit_ops

__all__ = ["SparseMultiHeadRMSNorm", "SparseMultiHeadAttention"]


class SparseMultiHeadRMSNorm(nn.Module):
    def __init__(self, dim: int, heads: int):
        super().__init__()
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.ones(heads, dim))

    def forward(self, x: Union[SparseTensor, torch.Tensor]) -> Union[SparseTensor, torch.Tensor]:
        x_type = x.dtype
        x = x.float()
        if isinstance(x, SparseTensor):
            x = x.replace(F.normalize(x.feats, dim=-1))
        else:
            x = F.normalize(x, dim=-1)
        return (x * self.gamma * self.scale).to(x_type)


class SparseMultiHeadAttention(nn.Module):
    def __init__(self, channels: int, num_heads: int, ctx_channels: Optional[int] = None,
                 type: Literal["self", "cross"] = "self",
                 attn_mode: Literal["full", "serialized", "windowed"] = "full", window_size: Optional[int] = None,
                 shift_sequence: Optional[int] = None, shift_window: Optional[Tuple[int, int, int]] = None,
                 serialize_mode: Optional[SerializeMode] = None, qkv_bias: bool = True, use_rope: bool = False,
                 qk_rms_norm: bool = False, use_old_attn_impl: bool = False):
        super().__init__()
        assert channels % num_heads == 0
        assert type in ["self", "cross"], f"Invalid attention type: {type}"
        assert attn_mode in ["full", "serialized", "windowed"], f"Invalid attention mode: {attn_mode}"
        assert type == "self" or attn_mode == "full", "Cross-attention only supports full attention"
        assert type == "self" or use_rope is False, "Rotary position embeddings only supported for self-attention"
        if use_rope:
            raise NotImplementedError("RoPE is not built")
        self.channels = channels
        self.ctx_channels = ctx_channels if ctx_channels is not None else channels
        self.num_heads = num_heads
        self._type = type
        self.attn_mode = attn_mode
        self.window_size = window_size
        self.shift_sequence = shift_sequence
        self.shift_window = shift_window
        self.serialize_mode = serialize_mode
        self.use_rope = use_rope
        self.qk_rms_norm = qk_rms_norm
        self.use_old_attn_impl = use_old_attn_impl
        if self._type == "self":
            self.to_qkv = nn.Linear(channels, channels * 3, bias=qkv_bias)
        else:
            self.to_q = nn.Linear(channels, channels, bias=qkv_bias)
            self.to_kv = nn.Linear(self.ctx_channels, channels * 2, bias=qkv_bias)
        if self.qk_rms_norm:
            self.q_rms_norm = SparseMultiHeadRMSNorm(channels // num_heads, num_heads)
            self.k_rms_norm = SparseMultiHeadRMSNorm(channels // num_heads, num_heads)
        self.to_out = nn.Linear(channels, channels)

    @staticmethod
    def _linear(lin: nn.Linear, x: torch.Tensor) -> torch.Tensor:
        """x (T, K) any float dtype -> (T, N) fp32 through the bf16 MFMA GEMM."""
        xb = dit_ops.cast_pad_bf16(x.float().contiguous(), dit_ops.pad64(x.shape[1]))
        wb = dit_ops.cast_pad_bf16(lin.weight.detach().float().contiguous(), dit_ops.pad64(x.shape[1]))
        out = torch.empty((x.shape[0], lin.out_features), dtype=torch.float32, device=x.device)
        bias = None if lin.bias is None else lin.bias.detach().float().contiguous()
        return dit_ops.gemm_bf16(xb, wb, bias, out, dit_ops.EPI_STORE_F32)

    def _project(self, lin, x):
        if isinstance(x, SparseTensor):
            return x.replace(self._linear(lin, x.feats).to(x.dtype))
        return self._linear(lin, x.reshape(-1, x.shape[-1])).reshape(*x.shape[:-1], -1).to(x.dtype)

    def _fused_pre(self, x, num_fused: int):
        """channels -> [num_fused, H, C]; the old implementation stored them [H, num_fused, C] (modules.py:150-162)."""
        H = self.num_heads
        f = x.feats.unsqueeze(0) if isinstance(x, SparseTensor) else x
        if self.use_old_attn_impl:
            f = torch.stack(f.reshape(*f.shape[:2], H, -1).chunk(num_fused, dim=-1), dim=2)
        else:
            f = f.reshape(*f.shape[:2], num_fused, H, -1)
        return x.replace(f.squeeze(0)) if isinstance(x, SparseTensor) else f

    def forward(self, x: Union[SparseTensor, torch.Tensor], context: Optional[Union[SparseTensor, torch.Tensor]] = None):
        H = self.num_heads
        if self._type == "self":
            qkv = self._fused_pre(self._project(self.to_qkv, x), 3)
            if self.qk_rms_norm:
                q, k, v = qkv.unbind(dim=1 if isinstance(qkv, SparseTensor) else 2)
                q, k = self.q_rms_norm(q), self.k_rms_norm(k)
                if isinstance(qkv, SparseTensor):
                    qkv = qkv.replace(torch.stack([q.feats, k.feats, v.feats], dim=1))
                else:
                    qkv = torch.stack([q, k, v], dim=2)
            if self.attn_mode == "full":
                h = sparse_scaled_dot_product_attention(qkv)
            elif self.attn_mode == "serialized":
                h = sparse_serialized_scaled_dot_product_self_attention(
                    qkv, self.window_size, serialize_mode=self.serialize_mode, shift_sequence=self.shift_sequence,
                    shift_window=self.shift_window)
            else:
                h = sparse_windowed_scaled_dot_product_self_attention(qkv, self.window_size, shift_window=self.shift_window)
        else:
            q = self._project(self.to_q, x)
            q = q.reshape(H, -1) if isinstance(q, SparseTensor) else q.reshape(*q.shape[:2], H, -1)
            kv = self._fused_pre(self._project(self.to_kv, context), 2)
            if self.qk_rms_norm:
                q = self.q_rms_norm(q)
                k, v = kv.unbind(dim=1 if isinstance(kv, SparseTensor) else 2)
                k = self.k_rms_norm(k)
                h = sparse_scaled_dot_product_attention(q, k, v)
            else:
                h = sparse_scaled_dot_product_attention(q, kv)
        h = h.reshape(-1) if isinstance(h, SparseTensor) else h.reshape(*h.shape[:2], -1)
        return self._project(self.to_out, h)
