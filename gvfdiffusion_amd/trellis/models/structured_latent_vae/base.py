"""SparseTransformerBase (trellis/models/structured_latent_vae/base.py:27-117): input layer + absolute position embedding
+ `num_blocks` sparse transformer blocks, on the row path of model/sparse_voxel_diffusion/sparse_transformer.py (same
block structure: TRELLIS' blocks are the unmodulated blocks of the static VAE with the [q|k|v][head][c] channel layout
and an optional QK-RMSNorm, which the attention kernel applies in its prologue)."""
from typing import *

import torch
import torch.nn as nn

from .... import sparse as sp
from ...._lib import require_cuda
from ....model.sparse_voxel_diffusion.sparse_transformer import AbsolutePositionEmbedder, SparseTransformerBlock, block_attn_config
from ....ops import dit_ops

__all__ = ["SparseTransformerBase"]


class SparseTransformerBase(nn.Module):
    def __init__(self, in_channels: int, model_channels: int, num_blocks: int, num_heads: Optional[int] = None,
                 num_head_channels: Optional[int] = 64, mlp_ratio: float = 4.0, attn_mode: str = "full",
                 window_size: Optional[int] = None, pe_mode: str = "ape", use_fp16: bool = False, use_checkpoint: bool = False,
                 qk_rms_norm: bool = False):
        super().__init__()
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.num_blocks = num_blocks
        self.window_size = window_size
        self.num_heads = num_heads or model_channels // num_head_channels
        self.mlp_ratio = mlp_ratio
        self.attn_mode = attn_mode
        self.pe_mode = pe_mode
        self.use_fp16 = use_fp16
        self.use_checkpoint = use_checkpoint
        self.qk_rms_norm = qk_rms_norm
        self.dtype = torch.float16 if use_fp16 else torch.float32
        if pe_mode == "ape":
            self.pos_embedder = AbsolutePositionEmbedder(model_channels)
        elif pe_mode == "rope":
            raise NotImplementedError("RoPE is not built")
        self.input_layer = sp.SparseLinear(in_channels, model_channels)
        self.blocks = nn.ModuleList([
            SparseTransformerBlock(model_channels, num_heads=self.num_heads, mlp_ratio=self.mlp_ratio, attn_mode=mode, window_size=ws,
                                   shift_sequence=shift_seq, shift_window=shift_win, serialize_mode=ser, use_checkpoint=use_checkpoint,
                                   modulated=False, use_rope=False, use_old_attn_impl=False, qk_rms_norm=qk_rms_norm)
            for mode, ws, shift_seq, shift_win, ser in block_attn_config(self)])

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    def convert_to_fp16(self) -> None:     # precision placement is inside the kernels
        pass

    def convert_to_fp32(self) -> None:
        pass

    def initialize_weights(self) -> None:
        for m in self.modules():
            if isinstance(m, nn.Linear):
                torch.nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    @torch.no_grad()
    def forward_rows(self, x: sp.SparseTensor) -> torch.Tensor:
        """-> the residual stream after the last block, fp32 (T, model_channels)."""
        require_cuda(x.feats, x.coords)
        lin = self.input_layer
        K = dit_ops.pad64(lin.in_features)
        T, C = x.feats.shape[0], self.model_channels
        h = self.pos_embedder(x.coords[:, 1:]).float().contiguous() if self.pe_mode == "ape" else torch.zeros((T, C), device=x.device)
        dit_ops.gemm_bf16(dit_ops.cast_pad_bf16(x.feats.float().contiguous(), K), dit_ops.cast_pad_bf16(lin.weight.detach().float().contiguous(), K),
                          lin.bias.detach().float().contiguous(), h, dit_ops.EPI_RESID_F32)
        for blk in self.blocks:
            blk.forward_rows(h, x)
        return h

    def forward(self, x: sp.SparseTensor) -> sp.SparseTensor:
        return x.replace(self.forward_rows(x).to(self.dtype))
