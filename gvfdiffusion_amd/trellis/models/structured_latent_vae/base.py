"""SparseTransformerBase -- the torso TRELLIS' structured-latent encoder / decoders share (trellis/models/structured_latent_vae/
base.py:27-117): `input_layer` + absolute position embedding + `blocks`, no output layer.  TRELLIS' block is the static VAE's
unmodulated block with the [q|k|v][head][c] channel layout and an optional QK-RMSNorm, so the torso is the row path of
gvfdiffusion_amd/model/sparse_voxel_diffusion/sparse_transformer.py (the attention kernel applies the RMS gains in its prologue)."""
from typing import Optional

import torch
import torch.nn as nn

from .... import sparse as sp
from ...._lib import require_cuda
from ....ops import precision
from ....model.sparse_voxel_diffusion.sparse_transformer import AbsolutePositionEmbedder, build_blocks, edge_weights, run_torso

__all__ = ["SparseTransformerBase"]


class SparseTransformerBase(nn.Module):
    def __init__(self, in_channels: int, model_channels: int, num_blocks: int, num_heads: Optional[int] = None,
                 num_head_channels: Optional[int] = 64, mlp_ratio: float = 4.0, attn_mode: str = "full",
                 window_size: Optional[int] = None, pe_mode: str = "ape", use_fp16: bool = False, use_checkpoint: bool = False,
                 qk_rms_norm: bool = False):
        super().__init__()
        if pe_mode not in ("ape", "rope"):
            raise ValueError(f"unknown pe_mode {pe_mode}")
        if pe_mode == "rope":
            raise NotImplementedError("RoPE is not built")
        cfg = dict(in_channels=in_channels, model_channels=model_channels, num_blocks=num_blocks, window_size=window_size,
                   num_heads=num_heads or model_channels // num_head_channels, mlp_ratio=mlp_ratio, attn_mode=attn_mode, pe_mode=pe_mode,
                   use_fp16=use_fp16, use_checkpoint=use_checkpoint, qk_rms_norm=qk_rms_norm,
                   dtype=torch.float16 if use_fp16 else torch.float32)         # the attributes upstream exposes
        for k, v in cfg.items():
            setattr(self, k, v)
        self.pos_embedder = AbsolutePositionEmbedder(model_channels)
        self.input_layer = sp.SparseLinear(in_channels, model_channels)
        self.blocks = build_blocks(self, model_channels, self.num_heads, mlp_ratio, use_checkpoint, qk_rms_norm=qk_rms_norm)
        self.compute_dtype = None

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    def set_compute_dtype(self, dtype):
        """torch.float16 / torch.bfloat16 (or "fp16" / "bf16"); None hands the choice back to ops/precision.py."""
        self.compute_dtype = precision.parse(dtype)
        return self

    def _lp(self):
        """16-bit operand type of the GEMMs / attention (fp32 accumulation and residual stream either way): fp16 when built with
        use_fp16 / after convert_to_fp16() -- upstream's torso then IS fp16 (base.py:93-101) --, bf16 otherwise."""
        return precision.resolve(self.compute_dtype, (), torch.float16 if self.use_fp16 else torch.bfloat16)

    def convert_to_fp16(self) -> None:
        """The parameters stay fp32 (cast once per version); from here on the kernels contract fp16 operands."""
        self.use_fp16, self.dtype = True, torch.float16

    def convert_to_fp32(self) -> None:
        self.use_fp16, self.dtype = False, torch.float32

    def initialize_weights(self) -> None:
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    @torch.no_grad()
    def forward_rows(self, x: sp.SparseTensor) -> torch.Tensor:
        """-> the residual stream after the last block, fp32 (T, model_channels)."""
        require_cuda(x.feats, x.coords)
        lp = self._lp()
        return run_torso(x, x.feats, edge_weights(self.input_layer, lp), self.pos_embedder, self.blocks, self.model_channels, lp)

    def forward(self, x: sp.SparseTensor) -> sp.SparseTensor:
        return x.replace(self.forward_rows(x).to(self.dtype))
