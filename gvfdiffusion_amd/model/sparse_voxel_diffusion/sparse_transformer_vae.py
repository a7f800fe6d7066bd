"""SparseTransformerVAE -- the static-VAE backbone (model/sparse_voxel_diffusion/sparse_transformer_vae.py:14-213) on
the MI355X kernels: same constructor keywords, same parameter tree (`input_layer`, `encoder.N.*`, `to_latent`,
`from_latent`, `decoder.N.*`, `out_layer`), `encode(x, sample_posterior, return_raw)`, `decode(latent)` and
`forward(x) -> (out, mean, logvar)`.

Precision placement (the reference casts the torso to fp16 with `use_fp16`, :164,192): the residual stream stays fp32
from the input layer to the output layer; every GEMM / attention operand is 16 bit with fp32 accumulation -- fp16 when the module was
built with `use_fp16=True` / after `convert_to_fp16()` (the reference's own arithmetic type), bf16 otherwise; `set_compute_dtype`,
`GVF_DIT_DTYPE` or an autocast region override it (ops/precision.py).  The elastic
memory controller (gradient checkpointing by memory ratio) is a training device and is accepted but ignored."""
from contextlib import contextmanager
from typing import *

import torch
import torch.nn as nn

from ... import sparse as sp
from ..._lib import require_cuda
from ...ops import dit_ops, precision
from .sparse_transformer import AbsolutePositionEmbedder, build_blocks, edge_weights, run_torso

__all__ = ["SparseTransformerVAE"]


class SparseTransformerVAE(nn.Module):
    def __init__(self, resolution, in_channels, model_channels, out_channels, latent_channels, num_blocks, window_size=1024,
                 num_heads=None, num_head_channels=64, mlp_ratio=4, attn_mode="swin", pe_mode="ape", use_fp16=False,
                 use_checkpoint=False, use_old_attn_impl=True, norm_output=False):
        super().__init__()
        self.resolution = resolution
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.latent_channels = latent_channels
        self.num_blocks = num_blocks
        self.window_size = window_size
        self.num_heads = num_heads or model_channels // num_head_channels
        self.mlp_ratio = mlp_ratio
        self.attn_mode = attn_mode
        self.pe_mode = pe_mode
        self.use_fp16 = use_fp16
        self.use_checkpoint = use_checkpoint
        self.norm_output = norm_output
        self.dtype = torch.float16 if use_fp16 else torch.float32
        if pe_mode == "ape":
            self.pos_embedder = AbsolutePositionEmbedder(model_channels)
        elif pe_mode == "rope":
            raise NotImplementedError("RoPE is not built (the released static VAE uses pe_mode='ape')")

        def blocks():
            return build_blocks(self, model_channels, self.num_heads, self.mlp_ratio, use_checkpoint, use_old_attn_impl=use_old_attn_impl)

        self.input_layer = sp.SparseLinear(in_channels, model_channels)
        self.encoder = blocks()
        self.to_latent = sp.SparseLinear(model_channels, 2 * latent_channels)
        self.from_latent = sp.SparseLinear(latent_channels, model_channels)
        self.decoder = blocks()
        self.out_layer = sp.SparseLinear(model_channels, out_channels)
        self.initialize_weights()
        self._wcache = None
        self.compute_dtype = None          # None: ops/precision.py's rule with the module default below

    @property
    def device(self):
        return next(self.parameters()).device

    def set_compute_dtype(self, dtype):
        """torch.float16 / torch.bfloat16 (or "fp16" / "bf16"); None hands the choice back to ops/precision.py."""
        self.compute_dtype = precision.parse(dtype)
        return self

    def _lp(self):
        return precision.resolve(self.compute_dtype, (), torch.float16 if self.use_fp16 else torch.bfloat16)

    def convert_to_fp16(self):   # the parameters stay fp32 (they are cast once per version); the kernels contract fp16
        self.use_fp16, self.dtype = True, torch.float16

    def convert_to_fp32(self):   # (no fp32 matrix path: 16-bit operands with fp32 accumulation, bf16 by default)
        self.use_fp16, self.dtype = False, torch.float32

    def initialize_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                torch.nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        for lin in (self.to_latent, self.out_layer):
            nn.init.constant_(lin.weight, 0)
            nn.init.constant_(lin.bias, 0)

    def freeze_encoder(self):
        for block in self.encoder:
            block.requires_grad_(False)

    # ---- weights of the four edge layers (bf16, K padded to 64) ----------------------------------------------------------
    def _weights(self):
        lp = self._lp()
        edge = (self.input_layer, self.to_latent, self.from_latent, self.out_layer)
        ver = (tuple((p.data_ptr(), p._version) for lin in edge for p in lin.parameters()), lp)
        if self._wcache is not None and self._wcache[0] == ver:
            return self._wcache[1]
        W = {name: edge_weights(lin, lp) for name, lin in zip(("input", "to_latent", "from_latent", "out"), edge)}
        self._wcache = (ver, W)
        return W

    def _torso(self, st: sp.SparseTensor, rows: torch.Tensor, w_in, blocks, w_out) -> torch.Tensor:
        """rows fp32 (T, K) -> edge GEMM (+ APE) -> blocks -> [LayerNorm] -> edge GEMM: fp32 (T, N_out)."""
        T, C = rows.shape[0], self.model_channels
        if T == 0:                                                                 # an empty voxel list maps to an empty one
            return torch.zeros((0, w_out[0].shape[0]), dtype=torch.float32, device=rows.device)
        lp = w_in[0].dtype
        x = run_torso(st, rows, w_in, self.pos_embedder if self.pe_mode == "ape" else None, blocks, C, lp)
        if self.norm_output:                                                       # F.layer_norm default eps (:166,194)
            hb = torch.empty((T, C), dtype=lp, device=rows.device)
            dit_ops.layernorm_modulate(x, hb, 1e-5)
        else:
            hb = dit_ops.cast_pad(x, C, dtype=lp)
        out = torch.empty((T, w_out[0].shape[0]), dtype=torch.float32, device=rows.device)
        return dit_ops.gemm(hb, *w_out, out, dit_ops.EPI_STORE_F32)

    @torch.no_grad()
    def encode(self, x: sp.SparseTensor, sample_posterior=True, return_raw=False):
        require_cuda(x.feats, x.coords)
        W = self._weights()
        h = self._torso(x, x.feats, W["input"], self.encoder, W["to_latent"])
        mean, logvar = h.chunk(2, dim=-1)
        z = mean + torch.exp(0.5 * logvar) * torch.randn_like(mean) if sample_posterior else mean
        z = x.replace(z.to(x.dtype))
        if return_raw:
            return z, mean.to(x.dtype), logvar.to(x.dtype)
        return z

    @torch.no_grad()
    def decode(self, latent: sp.SparseTensor) -> sp.SparseTensor:
        require_cuda(latent.feats, latent.coords)
        W = self._weights()
        return latent.replace(self._torso(latent, latent.feats, W["from_latent"], self.decoder, W["out"]).to(latent.dtype))

    def forward(self, x: sp.SparseTensor, t=None, c=None, mem_ratio=1.0):
        """-> (out, mean, logvar): ElasticModule.forward around _forward_with_mem_ratio (:206-210)."""
        latent, mean, logvar = self.encode(x, sample_posterior=True, return_raw=True)
        return self.decode(latent), mean, logvar

    def _get_input_size(self, x: sp.SparseTensor, t=None, c=None):
        return x.feats.shape[0]

    @contextmanager
    def with_mem_raio(self, mem_ratio=1.0):
        yield 1.0
