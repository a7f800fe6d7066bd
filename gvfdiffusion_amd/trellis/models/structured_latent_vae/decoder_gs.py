"""SLatGaussianDecoder (trellis/models/structured_latent_vae/decoder_gs.py:11-122): structured latent (8 channels per
active voxel) -> `num_gaussians` Gaussians per voxel.  forward = transformer torso -> LayerNorm -> out_layer ->
to_representation; the result objects are this package's GaussianModel (the accessors the renderer reads are the same in
TRELLIS' `Gaussian` twin)."""
from typing import *

import torch
import torch.nn as nn

from .... import sparse as sp
from ....model.sparse_voxel_diffusion.sparse_vae import hammersley_sequence
from ....ops import dit_ops
from ....representations.gaussian import Gaussian
from ....representations.gaussian.voxel_rows import gaussian_row_layout, rows_to_gaussian
from .base import SparseTransformerBase

__all__ = ["SLatGaussianDecoder"]


class SLatGaussianDecoder(SparseTransformerBase):
    def __init__(self, resolution: int, model_channels: int, latent_channels: int, num_blocks: int, num_heads: Optional[int] = None,
                 num_head_channels: Optional[int] = 64, mlp_ratio: float = 4, attn_mode: str = "swin", window_size: int = 8,
                 pe_mode: str = "ape", use_fp16: bool = False, use_checkpoint: bool = False, qk_rms_norm: bool = False,
                 representation_config: dict = None):
        super().__init__(in_channels=latent_channels, model_channels=model_channels, num_blocks=num_blocks, num_heads=num_heads,
                         num_head_channels=num_head_channels, mlp_ratio=mlp_ratio, attn_mode=attn_mode, window_size=window_size,
                         pe_mode=pe_mode, use_fp16=use_fp16, use_checkpoint=use_checkpoint, qk_rms_norm=qk_rms_norm)
        self.resolution = resolution
        self.rep_config = representation_config
        self._calc_layout()
        self.out_layer = sp.SparseLinear(model_channels, self.out_channels)
        self._build_perturbation()
        self.initialize_weights()

    def initialize_weights(self) -> None:
        super().initialize_weights()
        nn.init.constant_(self.out_layer.weight, 0)
        nn.init.constant_(self.out_layer.bias, 0)

    def _build_perturbation(self) -> None:
        n = self.rep_config["num_gaussians"]
        p = torch.tensor([hammersley_sequence(3, i, n) for i in range(n)]).float() * 2 - 1
        self.register_buffer("offset_perturbation", torch.atanh(p / self.rep_config["voxel_size"]))

    def _calc_layout(self) -> None:
        self.layout = gaussian_row_layout(self.rep_config["num_gaussians"])
        self.out_channels = self.layout["_opacity"]["range"][1]

    def to_representation(self, x: sp.SparseTensor) -> List[Gaussian]:
        cfg = self.rep_config
        kw = dict(mininum_kernel_size=cfg["3d_filter_kernel_size"], scaling_bias=cfg["scaling_bias"], opacity_bias=cfg["opacity_bias"],
                  scaling_activation=cfg["scaling_activation"])
        return [rows_to_gaussian(x.feats[sl], x.coords[sl][:, 1:], self.resolution, self.layout, cfg["lr"], 0.5 * cfg["voxel_size"],
                                 self.offset_perturbation if cfg["perturb_offset"] else None, kw) for sl in x.layout]

    @torch.no_grad()
    def decode_rows(self, x: sp.SparseTensor) -> sp.SparseTensor:
        """-> the (T, out_channels) output rows before to_representation."""
        if x.feats.shape[0] == 0:
            return x.replace(torch.zeros((0, self.out_channels), dtype=x.dtype, device=x.device))
        h = self.forward_rows(x)
        lp = self._lp()
        hb = torch.empty(h.shape, dtype=lp, device=h.device)
        dit_ops.layernorm_modulate(h, hb, 1e-5)                                   # F.layer_norm default eps (:119)
        lin = self.out_layer
        out = torch.empty((h.shape[0], self.out_channels), dtype=torch.float32, device=h.device)
        dit_ops.gemm(hb, lin.weight.detach().to(lp).contiguous(), lin.bias.detach().float().contiguous(), out, dit_ops.EPI_STORE_F32)
        return x.replace(out.to(x.dtype))

    def forward(self, x: sp.SparseTensor) -> List[Gaussian]:
        return self.to_representation(self.decode_rows(x))
