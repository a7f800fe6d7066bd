"""Sparse transformer blocks of the static VAE on the MI355X kernels
(model/sparse_voxel_diffusion/sparse_transformer.py:11-198: block_attn_config, AbsolutePositionEmbedder,
SparseFeedForward, SparseTransformerBlock -- same constructors and parameter names, so the released checkpoint's keys
load unchanged).

A block keeps the residual stream as fp32 rows (T, C) and runs, per sub-layer, the same kernels as the DiT:
LayerNorm -> 16-bit operand (bf16 or fp16: `lp`), MFMA GEMM with bias / tanh-GELU / residual epilogues, and the varlen flash attention
over the tokens gathered into window (or serialisation) order.  The partition of a SparseTensor is computed once per
(window, shift) on the device and cached on the tensor, including the device-side cu_seqlens."""
from typing import *

import numpy as np
import torch
import torch.nn as nn

from ... import sparse as sp
from ...ops import dit_ops, precision
from ...sparse.attention.modules import SparseMultiHeadAttention
from ...sparse.attention.serialized_attn import SerializeMode, SerializeModes, calc_serialization
from ...sparse.attention.windowed_attn import calc_window_partition

__all__ = ["block_attn_config", "AbsolutePositionEmbedder", "SparseFeedForward", "SparseTransformerBlock", "edge_weights", "run_torso", "build_blocks"]


def block_attn_config(self):
    """(attn_mode, window_size, shift_sequence, shift_window, serialize_mode) per block (:11-25)."""
    for i in range(self.num_blocks):
        if self.attn_mode == "shift_window":
            yield "serialized", self.window_size, 0, (16 * (i % 2),) * 3, SerializeMode.Z_ORDER
        elif self.attn_mode == "shift_sequence":
            yield "serialized", self.window_size, self.window_size // 2 * (i % 2), (0, 0, 0), SerializeMode.Z_ORDER
        elif self.attn_mode == "shift_order":
            yield "serialized", self.window_size, 0, (0, 0, 0), SerializeModes[i % 4]
        elif self.attn_mode == "full":
            yield "full", None, None, None, None
        elif self.attn_mode == "swin":
            yield "windowed", self.window_size, None, self.window_size // 2 * (i % 2), None


class AbsolutePositionEmbedder(nn.Module):
    """Per-axis sin / cos embedding of integer voxel coordinates, zero-padded to hidden_size (:73-112).  Step- and
    layer-invariant, (T, C) once per tensor: plain device arithmetic."""

    def __init__(self, hidden_size: int, in_channels: int = 3):
        super().__init__()
        self.hidden_size = hidden_size
        self.in_channels = in_channels
        self.freq_dim = hidden_size // in_channels // 2
        self.freqs = 1.0 / (10000 ** (torch.arange(self.freq_dim, dtype=torch.float32) / self.freq_dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        N, D = x.shape
        assert D == self.in_channels, "Input dimension must match number of input channels"
        self.freqs = self.freqs.to(x.device)
        out = torch.outer(x.reshape(-1).float(), self.freqs)
        embed = torch.cat([torch.sin(out), torch.cos(out)], dim=-1).reshape(N, -1)
        if embed.shape[1] < self.hidden_size:
            embed = torch.cat([embed, torch.zeros(N, self.hidden_size - embed.shape[1], device=embed.device)], dim=-1)
        return embed


class SparseFeedForward(nn.Module):
    def __init__(self, hidden_size: int, mlp_ratio: float = 4.0):
        super().__init__()
        self.mlp = nn.Sequential(sp.SparseLinear(hidden_size, int(hidden_size * mlp_ratio)), sp.SparseGELU(approximate="tanh"),
                                 sp.SparseLinear(int(hidden_size * mlp_ratio), hidden_size))

    def forward(self, x):
        return self.mlp(x)


def _lpw(w, lp):
    return w.detach().to(lp).contiguous()


def _fb(b):
    return None if b is None else b.detach().float().contiguous()


def token_partition(st: sp.SparseTensor, attn_mode: str, window_size, shift_sequence, shift_window, serialize_mode):
    """-> (fwd or None, bwd or None, cu_seqlens int32 on the device, longest sequence), cached on the tensor."""
    name = f"rows_{attn_mode}_{window_size}_{shift_sequence}_{shift_window}_{serialize_mode}"
    hit = st.get_spatial_cache(name)
    if hit is not None:
        return hit
    if attn_mode == "full":
        fwd = bwd = None
        lens = [s.stop - s.start for s in st.layout]
    elif attn_mode == "windowed":
        fwd, bwd, lens, _ = calc_window_partition(st, window_size, shift_window)
    else:
        fwd, bwd, lens, _ = calc_serialization(st, window_size, serialize_mode, shift_sequence, shift_window)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=st.device)
    hit = (fwd, bwd, cu, int(max(lens)))
    st.register_spatial_cache(name, hit)
    return hit


class SparseTransformerBlock(nn.Module):
    def __init__(self, hidden_size, num_heads, mlp_ratio=4.0, attn_mode="full", window_size=1024, shift_sequence=0,
                 shift_window=(0, 0, 0), serialize_mode=SerializeMode.Z_ORDER, use_checkpoint=False, modulated=True,
                 use_rope=False, use_old_attn_impl=False, qk_rms_norm=False):
        super().__init__()
        if modulated:
            raise NotImplementedError("adaLN-modulated sparse blocks belong to the spconv flow models (out of scope, DESIGN.md section 7)")
        self.use_checkpoint = use_checkpoint
        self.norm1 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.attn = SparseMultiHeadAttention(hidden_size, num_heads=num_heads, attn_mode=attn_mode, window_size=window_size,
                                             shift_sequence=shift_sequence, shift_window=shift_window, qkv_bias=True,
                                             serialize_mode=serialize_mode, use_rope=use_rope, use_old_attn_impl=use_old_attn_impl,
                                             qk_rms_norm=qk_rms_norm)
        self.norm2 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.mlp = SparseFeedForward(hidden_size, mlp_ratio=mlp_ratio)
        self.modulated = modulated
        self._wcache = None

    def _weights(self, lp=torch.bfloat16):
        ver = (tuple((p.data_ptr(), p._version) for p in self.parameters()), lp)
        if self._wcache is not None and self._wcache[0] == ver:
            return self._wcache[1]
        _bf = lambda w: _lpw(w, lp)
        a = self.attn
        C, H = a.channels, a.num_heads
        wq, bq = a.to_qkv.weight.detach(), a.to_qkv.bias.detach()
        if a.use_old_attn_impl:      # rows ordered [head][q|k|v][c] (sparse/attention/modules.py:150-158) -> [q|k|v][head][c]
            wq = wq.reshape(H, 3, C // H, C).permute(1, 0, 2, 3).reshape(3 * C, C)
            bq = bq.reshape(H, 3, C // H).permute(1, 0, 2).reshape(3 * C)
        W = dict(qkv=(_bf(wq), _fb(bq)), out=(_bf(a.to_out.weight), _fb(a.to_out.bias)),
                 gq=a.q_rms_norm.gamma.detach().float().contiguous() if a.qk_rms_norm else None,
                 gk=a.k_rms_norm.gamma.detach().float().contiguous() if a.qk_rms_norm else None,
                 fc1=(_bf(self.mlp.mlp[0].weight), _fb(self.mlp.mlp[0].bias)), fc2=(_bf(self.mlp.mlp[2].weight), _fb(self.mlp.mlp[2].bias)))
        self._wcache = (ver, W)
        return W

    @torch.no_grad()
    def forward_rows(self, x: torch.Tensor, st: sp.SparseTensor, lp=None) -> torch.Tensor:
        """x: fp32 (T, C) residual rows of `st` (updated in place and returned).  lp: the 16-bit operand type of the GEMMs and the attention
        (torch.float16 / torch.bfloat16; None: ops/precision.py's rule -- environment, autocast region, else bf16)."""
        lp = precision.resolve(lp, (), torch.bfloat16)
        W = self._weights(lp)
        a = self.attn
        T, C = x.shape
        H = a.num_heads
        d = C // H
        bf16, dev = lp, x.device
        fwd, bwd, cu, longest = token_partition(st, a.attn_mode, a.window_size, a.shift_sequence, a.shift_window, a.serialize_mode)
        hb = torch.empty((T, C), dtype=bf16, device=dev)
        dit_ops.layernorm_modulate(x, hb, self.norm1.eps)
        qkv = torch.empty((T, 3 * C), dtype=bf16, device=dev)
        dit_ops.gemm(hb, *W["qkv"], qkv, dit_ops.EPI_STORE_16)
        g = qkv if fwd is None else qkv.index_select(0, fwd)                      # (M, 3C) in sequence order
        ao = torch.empty((g.shape[0], C), dtype=bf16, device=dev)
        s3 = (0, 0, 3 * C)
        dit_ops.attention_varlen(g, g[:, C:], g[:, 2 * C:], ao, cu, cu, longest, longest, H, s3, s3, s3, (0, 0, C),
                                 W["gq"], W["gk"], head_dim=d)
        if bwd is not None:
            ao = ao.index_select(0, bwd)
        dit_ops.gemm(ao, *W["out"], x, dit_ops.EPI_RESID_F32)
        dit_ops.layernorm_modulate(x, hb, self.norm2.eps)
        hid = torch.empty((T, W["fc1"][0].shape[0]), dtype=bf16, device=dev)
        dit_ops.gemm(hb, *W["fc1"], hid, dit_ops.EPI_GELU_16)
        dit_ops.gemm(hid, *W["fc2"], x, dit_ops.EPI_RESID_F32)
        return x

    def forward(self, x: sp.SparseTensor, c: torch.Tensor = None) -> sp.SparseTensor:
        rows = x.feats.float().contiguous().clone()
        return x.replace(self.forward_rows(rows, x, precision.resolve(None, (x.feats,), torch.bfloat16)).to(x.dtype))


def edge_weights(lin: nn.Linear, lp=torch.bfloat16):
    """(weight as `lp` (bf16 / fp16) with K zero-padded to a multiple of 64, fp32 bias) of an input / output layer."""
    return (dit_ops.cast_pad(lin.weight.detach().float().contiguous(), dit_ops.pad64(lin.in_features), dtype=lp),
            lin.bias.detach().float().contiguous())


@torch.no_grad()
def run_torso(st: sp.SparseTensor, rows: torch.Tensor, w_in, pos_embedder, blocks, channels: int, lp=None) -> torch.Tensor:
    """rows (T, K) of `st` -> fp32 residual stream (T, channels) after  rows @ W_in^T + b (+ position embedding)  and every block.
    The embedding is written first and the input GEMM accumulates onto it (residual epilogue)."""
    T = rows.shape[0]
    if pos_embedder is not None:
        x = pos_embedder(st.coords[:, 1:]).float().contiguous()
    else:
        x = torch.zeros((T, channels), dtype=torch.float32, device=rows.device)
    lp = w_in[0].dtype if lp is None else lp                     # the edge weights were cast to the operand type of this pass
    a = dit_ops.cast_pad(rows.float().contiguous(), w_in[0].shape[1], dtype=lp)
    dit_ops.gemm(a, *w_in, x, dit_ops.EPI_RESID_F32)
    for blk in blocks:
        blk.forward_rows(x, st, lp)
    return x


def build_blocks(owner, channels: int, num_heads: int, mlp_ratio: float, use_checkpoint: bool = False, **block_kw) -> nn.ModuleList:
    """One unmodulated SparseTransformerBlock per entry of block_attn_config(owner) (owner: num_blocks, attn_mode, window_size)."""
    return nn.ModuleList([
        SparseTransformerBlock(channels, num_heads=num_heads, mlp_ratio=mlp_ratio, attn_mode=mode, window_size=window,
                               shift_sequence=shift_seq, shift_window=shift_win, serialize_mode=order, use_checkpoint=use_checkpoint,
                               modulated=False, use_rope=False, **block_kw)
        for mode, window, shift_seq, shift_win, order in block_attn_config(owner)])
