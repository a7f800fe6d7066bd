"""MultiHeadRMSNorm / MultiHeadAttention with the reference's constructor, parameter names and
forward contract (model/attention/modules.py:8-15,63-146), computed by the gfx950 kernels:
fp16 / bf16 MFMA projections (csrc/gemm.hip; the type: ops/precision.py) and flash attention with the QK-RMSNorm fused into its operand
loads (csrc/attn.hip).  RoPE is not built: DiT passes use_rope=(pe_mode == "rope") (model/dit.py:376; configs/diffusion.yml: "ape"), and the
reference's RotaryPositionEmbedder cannot run on the dense (B, L, H, d) tensors this module would hand it -- its default indices index the
HEAD axis and its (B, L, H * freq_dim) phases do not broadcast against the (B, L, H, d / 2) complex view (dead code upstream)."""
from typing import *

import torch
import torch.nn as nn

from ...ops import dit_ops, precision

__all__ = ["MultiHeadRMSNorm", "MultiHeadAttention"]


class MultiHeadRMSNorm(nn.Module):
    """x <- normalize(x.float(), dim=-1) * gamma[H, d] * sqrt(d).  Inside MultiHeadAttention the gain is
    handed to the attention kernel; called on its own it runs the same formula with torch ops."""

    def __init__(self, dim: int, heads: int):
        super().__init__()
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.ones(heads, dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return (torch.nn.functional.normalize(x.float(), dim=-1) * self.gamma * self.scale).to(x.dtype)


class _WeightCache:
    """16-bit (K padded to a multiple of 64) copies of nn.Linear weights, refreshed when the parameter changes."""

    def __init__(self):
        self._c = {}

    def get(self, lin: nn.Linear, lp=torch.bfloat16):
        w = lin.weight
        key = (id(lin), lp)
        ver = (w._version, w.data_ptr(), w.device)
        hit = self._c.get(key)
        if hit is None or hit[0] != ver:
            k = w.shape[1]
            wb = dit_ops.cast_pad(w.detach().float().contiguous(), dit_ops.pad64(k), dtype=lp)
            b = None if lin.bias is None else lin.bias.detach().float().contiguous()
            hit = (ver, wb, b)
            self._c[key] = hit
        return hit[1], hit[2]


class MultiHeadAttention(nn.Module):
    def __init__(self, channels: int, num_heads: int, ctx_channels: Optional[int] = None,
                 type: Literal["self", "cross"] = "self", attn_mode: Literal["full", "windowed"] = "full",
                 window_size: Optional[int] = None, shift_window: Optional[Tuple[int, int, int]] = None,
                 qkv_bias: bool = True, use_rope: bool = False, qk_rms_norm: bool = False):
        super().__init__()
        assert channels % num_heads == 0
        assert type in ["self", "cross"], f"Invalid attention type: {type}"
        assert attn_mode in ["full", "windowed"], f"Invalid attention mode: {attn_mode}"
        assert type == "self" or attn_mode == "full", "Cross-attention only supports full attention"
        if attn_mode == "windowed":
            raise NotImplementedError("Windowed attention is not yet implemented")
        if use_rope:
            raise NotImplementedError("RoPE is not built (configs/diffusion.yml uses pe_mode 'ape')")
        self.channels = channels
        self.head_dim = channels // num_heads
        self.ctx_channels = ctx_channels if ctx_channels is not None else channels
        self.num_heads = num_heads
        self._type = type
        self.attn_mode = attn_mode
        self.window_size = window_size
        self.shift_window = shift_window
        self.use_rope = use_rope
        self.qk_rms_norm = qk_rms_norm
        if self._type == "self":
            self.to_qkv = nn.Linear(channels, channels * 3, bias=qkv_bias)
        else:
            self.to_q = nn.Linear(channels, channels, bias=qkv_bias)
            self.to_kv = nn.Linear(self.ctx_channels, channels * 2, bias=qkv_bias)
        if self.qk_rms_norm:
            self.q_rms_norm = MultiHeadRMSNorm(self.head_dim, num_heads)
            self.k_rms_norm = MultiHeadRMSNorm(self.head_dim, num_heads)
        self.to_out = nn.Linear(channels, channels)
        self._wc = _WeightCache()
        self.compute_dtype = None          # None: ops/precision.py decides per call (input type, autocast, bf16)

    def set_compute_dtype(self, dtype):
        self.compute_dtype = precision.parse(dtype)
        return self

    def _gammas(self):
        if not self.qk_rms_norm:
            return None, None
        return self.q_rms_norm.gamma.detach().float().contiguous(), self.k_rms_norm.gamma.detach().float().contiguous()

    def _proj(self, x2d, lin, out_dtype=None):
        w, b = self._wc.get(lin, x2d.dtype)
        out_dtype = x2d.dtype if out_dtype is None else out_dtype
        out = torch.empty((x2d.shape[0], lin.out_features), dtype=out_dtype, device=x2d.device)
        epi = dit_ops.EPI_STORE_F32 if out_dtype == torch.float32 else dit_ops.EPI_STORE_16
        return dit_ops.gemm(x2d, w, b, out, epi)

    def forward(self, x: torch.Tensor, context: Optional[torch.Tensor] = None, indices: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, L, C = x.shape
        H, d = self.num_heads, self.head_dim
        lp = precision.resolve(self.compute_dtype, (x, context))
        xb = dit_ops.cast_pad(x.reshape(B * L, C).float().contiguous(), dit_ops.pad64(C), dtype=lp)
        gq, gk = self._gammas()
        attn = torch.empty((B * L, C), dtype=lp, device=x.device)
        if self._type == "self":
            qkv = self._proj(xb, self.to_qkv)                      # (B*L, 3C) = [q | k | v] per token
            s = (L * 3 * C, 0, 3 * C)
            dit_ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], attn, B, 1, L, L, H, s, s, s, (L * C, 0, C), gq, gk, head_dim=d)
        else:
            Lkv = context.shape[1]
            cb = dit_ops.cast_pad(context.reshape(B * Lkv, -1).float().contiguous(), dit_ops.pad64(context.shape[-1]), dtype=lp)
            q = self._proj(xb, self.to_q)
            kv = self._proj(cb, self.to_kv)                        # (B*Lkv, 2C) = [k | v]
            sk = (Lkv * 2 * C, 0, 2 * C)
            dit_ops.attention(q, kv, kv[:, C:], attn, B, 1, L, Lkv, H, (L * C, 0, C), sk, sk, (L * C, 0, C), gq, gk, head_dim=d)
        out = self._proj(attn, self.to_out, out_dtype=torch.float32)
        return out.reshape(B, L, C).to(x.dtype)
