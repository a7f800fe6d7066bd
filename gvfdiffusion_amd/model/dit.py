"""Temporal-aware DiT with the reference's module surface (model/dit.py:16-480): same constructor
keywords, same parameter names (the 446-tensor state_dict of configs/diffusion.yml loads unchanged),
same forward(x, t, cond_images, static_latent, deformation_position_xyz) contract.

The forward pass does not call the sub-modules one by one: it drives the gfx950 kernels directly
(csrc/gemm.hip, attn.hip, elem.hip through ops/dit_ops.py) so that
  * LayerNorm + adaLN modulate / affine is one kernel that writes the bf16 GEMM operand,
  * gate * h + residual is the epilogue of the to_out / mlp.2 GEMMs on the fp32 residual stream,
  * GELU-tanh is the epilogue of mlp.0, QK-RMSNorm lives in the attention operand loads,
  * the (B,T,N,C) <-> (B,N,T,C) transposes of the temporal attention are strides, not copies,
  * all 25 adaLN projections of a step are ONE GEMM over a concatenated weight,
  * everything that depends only on the conditions -- image_cond_proj, static_cond_proj, every block's
    to_kv(context), the position embedding -- is computed once per condition set and reused for all
    NFEs (static K/V additionally once per sample instead of once per frame: dit.py:465 repeats it over T).
Numerics follow torch.autocast placement: 16-bit contraction operands, fp32 accumulation, fp32 residual
stream / LayerNorm / softmax / modulation.  The operand type is fp16 -- what the reference runs (accelerate
mixed_precision='fp16', inference_dpm_latent.py:122-125; configs/diffusion.yml use_fp16: true) -- or bf16 (BASELINE.json),
same MFMA rate: ops/precision.py has the rule, `set_compute_dtype` / GVF_DIT_DTYPE the override.
There is no CPU path: CPU tensors raise (the fp32 CPU restatement lives in oracle/dit_ref.py, tests only).
"""
import math
import os
from typing import *

import numpy as np
import torch
import torch.nn as nn

from .attention import MultiHeadAttention
from ..ops import dit_ops, precision
from .. import _lib


_CAPTURE_LOCK = __import__("threading").Lock()      # hipGraph captures of DiT instances are serialised (see DiT._forward_graphed)
_CTX_EPOCHS = __import__("itertools").count(1)      # names of the condition cache's buffer sets (DiT.prepare_conditions): a captured graph is keyed on one


class AbsolutePositionEmbedder(nn.Module):
    """(B, L, in_channels) positions -> (B, L, channels): per axis [sin(x f_i), cos(x f_i)], zero-padded
    (model/dit.py:16-56).  Step-invariant, evaluated once per condition set with torch device ops."""

    def __init__(self, channels: int, in_channels: int = 3):
        super().__init__()
        self.channels = channels
        self.in_channels = in_channels
        self.freq_dim = channels // in_channels // 2
        freqs = torch.arange(self.freq_dim, dtype=torch.float32) / self.freq_dim
        self.freqs = 1.0 / (10000 ** freqs)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, L, D = x.shape
        assert D == self.in_channels, "Input dimension must match number of input channels"
        self.freqs = self.freqs.to(x.device)
        out = torch.outer(x.reshape(-1).float(), self.freqs)
        emb = torch.cat([torch.sin(out), torch.cos(out)], dim=-1).reshape(B * L, -1)
        if emb.shape[1] < self.channels:
            emb = torch.cat([emb, torch.zeros(B * L, self.channels - emb.shape[1], device=emb.device)], dim=-1)
        return emb.reshape(B, L, -1)


class TimestepEmbedder(nn.Module):
    """[cos | sin](t * 10000^(-i/128)) -> Linear -> SiLU -> Linear (model/dit.py:59-100)."""

    def __init__(self, hidden_size, frequency_embedding_size=256):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(frequency_embedding_size, hidden_size, bias=True), nn.SiLU(),
                                 nn.Linear(hidden_size, hidden_size, bias=True))
        self.frequency_embedding_size = frequency_embedding_size

    @staticmethod
    def timestep_embedding(t, dim, max_period=10000):
        half = dim // 2
        freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
        args = t[:, None].float() * freqs[None]
        emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        if dim % 2:
            emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
        return emb


class FeedForwardNet(nn.Module):
    def __init__(self, channels: int, mlp_ratio: float = 4.0):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(channels, int(channels * mlp_ratio)), nn.GELU(approximate="tanh"),
                                 nn.Linear(int(channels * mlp_ratio), channels))


class ModulatedSparseTransformerCrossBlock(nn.Module):
    """Parameter container of one block (model/dit.py:141-225); executed by DiT.forward."""

    def __init__(self, channels: int, ctx_channels: int, num_heads: int, mlp_ratio: float = 4.0, attn_mode="full",
                 window_size=None, shift_sequence=None, shift_window=None, serialize_mode=None, use_checkpoint=False,
                 use_rope=False, qk_rms_norm=False, qk_rms_norm_cross=False, qkv_bias=True, share_mod=False,
                 no_temporal_attn=False):
        super().__init__()
        self.use_checkpoint = use_checkpoint
        self.share_mod = share_mod
        self.no_temporal_attn = no_temporal_attn
        self.norm1 = nn.LayerNorm(channels, elementwise_affine=False, eps=1e-6)
        self.norm2 = nn.LayerNorm(channels, elementwise_affine=False, eps=1e-6) if not no_temporal_attn else nn.Identity()
        self.norm3 = nn.LayerNorm(channels, elementwise_affine=True, eps=1e-6)
        self.norm4 = nn.LayerNorm(channels, elementwise_affine=True, eps=1e-6)
        self.norm5 = nn.LayerNorm(channels, elementwise_affine=False, eps=1e-6)
        mk = dict(num_heads=num_heads, qkv_bias=qkv_bias)
        self.spatial_self_attn = MultiHeadAttention(channels, type="self", attn_mode=attn_mode, window_size=window_size,
                                                    shift_window=shift_window, use_rope=use_rope, qk_rms_norm=qk_rms_norm, **mk)
        self.temporal_self_attn = MultiHeadAttention(channels, type="self", attn_mode=attn_mode, window_size=window_size,
                                                     shift_window=shift_window, use_rope=use_rope, qk_rms_norm=qk_rms_norm,
                                                     **mk) if not no_temporal_attn else nn.Identity()
        self.image_cross_attn = MultiHeadAttention(channels, ctx_channels=ctx_channels, type="cross", attn_mode="full",
                                                   qk_rms_norm=qk_rms_norm_cross, **mk)
        self.static_cross_attn = MultiHeadAttention(channels, ctx_channels=ctx_channels, type="cross", attn_mode="full",
                                                    qk_rms_norm=qk_rms_norm_cross, **mk)
        self.mlp = FeedForwardNet(channels, mlp_ratio=mlp_ratio)
        if not share_mod:
            self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(channels, 6 * channels, bias=True))
            self.adaLN_modulation_temporal = nn.Sequential(nn.SiLU(), nn.Linear(channels, 3 * channels, bias=True)) \
                if not no_temporal_attn else nn.Identity()


class FinalLayer(nn.Module):
    def __init__(self, hidden_size, out_channels):
        super().__init__()
        self.norm_final = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.linear = nn.Linear(hidden_size, out_channels, bias=True)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 2 * hidden_size, bias=True))


class DiT(nn.Module):
    def __init__(self, resolution: int, in_channels: int, model_channels: int, static_cond_channels: int,
                 image_cond_channels: int, out_channels: int, num_blocks: int, num_heads: Optional[int] = None,
                 num_head_channels: Optional[int] = 64, mlp_ratio: float = 4, patch_size: int = 1,
                 pe_mode: Literal["ape", "rope", "learnable", "none"] = "learnable", use_fp16: bool = False,
                 use_checkpoint: bool = False, use_skip_connection: bool = True, share_mod: bool = False,
                 qk_rms_norm: bool = False, qk_rms_norm_cross: bool = False, no_temporal_attn: bool = True):
        super().__init__()
        self.resolution = resolution
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.static_cond_channels = static_cond_channels
        self.image_cond_channels = image_cond_channels
        self.out_channels = out_channels
        self.num_blocks = num_blocks
        self.num_heads = num_heads or model_channels // num_head_channels
        self.mlp_ratio = mlp_ratio
        self.patch_size = patch_size
        self.pe_mode = pe_mode
        self.use_fp16 = use_fp16
        self.use_checkpoint = use_checkpoint
        self.use_skip_connection = use_skip_connection
        self.share_mod = share_mod
        self.qk_rms_norm = qk_rms_norm
        self.qk_rms_norm_cross = qk_rms_norm_cross
        self.dtype = torch.float16 if use_fp16 else torch.float32
        self.no_temporal_attn = no_temporal_attn
        assert int(np.log2(patch_size)) == np.log2(patch_size), "Patch size must be a power of 2"
        if share_mod:
            # the reference cannot run this option either: its shared projection is 6C wide where the block splits it 9 ways (and 9C / 6 ways
            # without temporal attention), model/dit.py:234-238 against :354-358 -> RuntimeError at :247 (scripts/reference_dit_variants.py,
            # profiles/r04_reference_dit_variants.txt).  There is no behaviour to reproduce.
            raise NotImplementedError("share_mod=True: the reference's own forward fails with a shape error at model/dit.py:247 "
                                      "(profiles/r04_reference_dit_variants.txt); configs/diffusion.yml uses per-block adaLN")
        if pe_mode == "rope":
            # dead in the reference as well: RotaryPositionEmbedder builds hidden_size // 3 // 2 = 5 phases per token for a 16-pair head and pads
            # along the token axis (model/attention/modules.py:36, 52-57) -> RuntimeError for every sequence length
            raise NotImplementedError("pe_mode='rope': the reference's own forward fails with a shape error at model/attention/modules.py:36 "
                                      "(profiles/r04_reference_dit_variants.txt); configs/diffusion.yml uses 'ape'")
        if model_channels % self.num_heads != 0 or model_channels // self.num_heads not in (32, 64):
            # head_dim 32 (configs/diffusion.yml: 512 channels / 16 heads) runs the tiled K / V cache (csrc/attn_xt.hip) and the row-block
            # launches; head_dim 64 (round 6; the reference takes any num_heads, model/dit.py:337) runs the per-sub-layer launches with the
            # strided flash attention of csrc/attn.hip, which is built for 32 and 64.  Anything else has no HIP attention path.
            raise NotImplementedError(f"the DiT's HIP attention paths are built for head_dim 32 and 64, got {model_channels} / {self.num_heads}")
        self.head_dim = model_channels // self.num_heads

        self.t_embedder = TimestepEmbedder(model_channels)
        if pe_mode == "ape":
            self.pos_embedder = AbsolutePositionEmbedder(model_channels)
        elif pe_mode == "learnable":
            self.pos_embedder = nn.Parameter(torch.randn(1, resolution, model_channels))
        self.input_layer = nn.Linear(in_channels, model_channels)
        self.blocks = nn.ModuleList([
            ModulatedSparseTransformerCrossBlock(model_channels, model_channels, num_heads=self.num_heads,
                                                 mlp_ratio=self.mlp_ratio, attn_mode="full",
                                                 use_checkpoint=self.use_checkpoint, use_rope=False,
                                                 share_mod=self.share_mod, qk_rms_norm=self.qk_rms_norm,
                                                 qk_rms_norm_cross=self.qk_rms_norm_cross,
                                                 no_temporal_attn=self.no_temporal_attn)
            for _ in range(num_blocks)])
        self.final_layer = FinalLayer(model_channels, out_channels)
        self.static_cond_proj = nn.Linear(static_cond_channels, model_channels)
        self.image_cond_proj = nn.Linear(image_cond_channels, model_channels)
        self.initialize_weights()
        self._wcache = None       # 16-bit weights, rebuilt when any parameter (or the compute dtype) changes
        self.compute_dtype = None # None: ops/precision.py decides per forward (GVF_DIT_DTYPE, autocast, use_fp16 -> fp16 else bf16)
        self._ctx_cache = {}      # step-invariant condition products
        self.use_graph = False    # replay the whole forward as one hipGraph (set by enable_graph)
        import os
        # LayerNorm folded into the GEMMs (see _forward): 0 = off (default), 1 = every projection, 2 = only the narrow ones
        # (N <= 512).  Built, bit-checked (tests/test_dit_gpu.py::test_layernorm_folded_into_the_gemms) and measured on the
        # denoise step: 7.35-7.49 / 7.67-7.91 / 7.35-7.48 ms per NFE for 0 / 1 / 2 -- the register-staged, re-normalised A
        # operand costs the wide projections (12-16 column tiles re-normalise the same rows) more than the LayerNorm pass it
        # removes (to_qkv 36.6 + 9.1 -> 50.7 us, mlp.0 50 + 9.1 -> 64.4 us; to_q 15.8 + 9.1 -> 20.7 us), and the statistics
        # add 2.4 us to every residual GEMM.
        self.fuse_layernorm = int(os.environ.get("GVF_DIT_FUSE_LN", "0"))
        # one launch per sub-layer boundary (csrc/rowblock.hip) where the shapes allow it: 0 = the unfused GEMM / LayerNorm launches
        self.use_rowblock = int(os.environ.get("GVF_DIT_ROWBLOCK", "1")) != 0
        self.weight_prefetch = int(os.environ.get("GVF_DIT_PREFETCH", "1")) != 0
        # precompute_modulation: on (GVF_DIT_MODTABLE=0 switches it off).  4.84-4.86 ms per step with the table against 4.88-4.89 without (two alternating
        # repeats on one box).  It first measured 0.35 ms SLOWER (profiles/r04_modtable_ab.txt) -- under the sampler that still ran in lock-step with
        # the device, which paid the table's host-side lookup on the critical path.
        self.modulation_table = int(os.environ.get("GVF_DIT_MODTABLE", "1")) != 0
        self._fallbacks = None         # device int32 (1,): count_attention_fallbacks()
        self.rowblock_tiled_kv = int(os.environ.get("GVF_DIT_TILED_KV", "1")) != 0    # to_qkv's launch writes the attention's K / V^T tiles itself
        # the temporal self attention runs INSIDE the row-block launch between the spatial and the image attention (T | 48): 6 launches per
        # block instead of 8, the qkv / attention-output buffers of the temporal sub-layer never exist
        self.rowblock_temporal = int(os.environ.get("GVF_DIT_TEMPORAL_FUSED", "1")) != 0
        self._graph = None
        self.capture_blocks = None     # diagnostics: a list -> _blocks_rowblock appends a copy of the fp32 stream after every block

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    def set_compute_dtype(self, dtype):
        """torch.float16 / torch.bfloat16 (or "fp16" / "bf16"); None hands the choice back to ops/precision.py."""
        self.compute_dtype = precision.parse(dtype)
        return self

    def _lp(self):
        """The matrix pipe's operand type for this forward (see ops/precision.py).  The inputs are fp32 in the reference's
        pipeline; a 16-bit x does not change the choice (the stream is fp32 either way)."""
        return precision.resolve(self.compute_dtype, (), torch.float16 if self.use_fp16 else torch.bfloat16)

    def initialize_weights(self) -> None:
        """Same scheme as model/dit.py:401-427 (xavier Linear, zero bias, N(0,0.02) embedders, zero adaLN / head)."""
        def _basic_init(module):
            if isinstance(module, nn.Linear):
                torch.nn.init.xavier_uniform_(module.weight)
                if module.bias is not None:
                    nn.init.constant_(module.bias, 0)
        self.apply(_basic_init)
        nn.init.normal_(self.t_embedder.mlp[0].weight, std=0.02)
        nn.init.normal_(self.t_embedder.mlp[2].weight, std=0.02)
        nn.init.normal_(self.static_cond_proj.weight, std=0.02)
        nn.init.normal_(self.image_cond_proj.weight, std=0.02)
        for block in self.blocks:
            nn.init.constant_(block.adaLN_modulation[-1].weight, 0)
            nn.init.constant_(block.adaLN_modulation[-1].bias, 0)
        nn.init.constant_(self.final_layer.adaLN_modulation[-1].weight, 0)
        nn.init.constant_(self.final_layer.adaLN_modulation[-1].bias, 0)
        nn.init.constant_(self.final_layer.linear.weight, 0)
        nn.init.constant_(self.final_layer.linear.bias, 0)

    # ---- weight preparation ---------------------------------------------------------------------
    def _param_version(self):
        """(version counter, storage address) of every parameter: what the packed-weight caches and the captured graph are keyed on.  Asked for on
        every forward, so the walk over the module tree (1.4 ms per call, a quarter of a denoise step once the sampler no longer waits for the
        device) is done once: the `_modules` / `_parameters` dicts of the tree are kept and read each time.  An in-place update, a .to() / .half(),
        a load_state_dict and a Parameter assigned to an existing slot show up in the (version, address) pairs; a replaced, added or removed
        SUBMODULE, a newly registered parameter and a parameter going from None to a tensor show up in the structural fingerprint (identity of
        every child in every kept `_modules` dict + the size of every `_parameters` dict; ~30 us), which sends the call back to the full walk."""
        d = self.__dict__
        held = d.get("_pver_held")
        if held is not None:
            return held
        st = d.get("_pstate")
        if st is not None:
            mdicts, pdicts, kids, sizes = st
            # `kids` are the child MODULES themselves (strong references), compared by identity: a bare id() would let a child that was deleted
            # and replaced by a new module at the recycled address pass the check with stale `_parameters` dicts
            now = [c for md in mdicts for c in md.values()]
            if len(now) != len(kids) or any(a is not b for a, b in zip(now, kids)) or sizes != tuple(len(pd) for pd in pdicts):
                st = None
        if st is None:
            mods = list(self.modules())
            mdicts, pdicts = [m._modules for m in mods], [m._parameters for m in mods]
            d["_pstate"] = (mdicts, pdicts, [c for md in mdicts for c in md.values()], tuple(len(pd) for pd in pdicts))
            # every structural change is a new key, whatever the allocator does: a replaced child's fresh parameters may land on the old one's
            # addresses with version 0, i.e. with the same (version, address) pairs and other VALUES
            d["_pstruct_epoch"] = d.get("_pstruct_epoch", -1) + 1
        return (("structure", d["_pstruct_epoch"]),) + tuple((-1, 0) if q is None else (q._version, q.data_ptr()) for pd in pdicts for q in pd.values())

    def hold_param_version(self, active: bool):
        """A sampler brackets ONE sample() call with hold_param_version(True) / (False) (DPM_Solver.sample through model_wrapper's
        `sampling_scope`): nobody updates weights inside a sampling call, so the parameter walk is done once for all its evaluations instead of
        once per forward.  It is host time, and exposed exactly where the sampler waits for the device -- the adaptive solver's step-size test
        empties the queue 22 times per sample, and the next launch then waits for the walk (~0.15 ms).
        CONTRACT: while a hold is active, a weight change (an optimizer / EMA step, load_state_dict, a `correcting_xt_fn` hook that writes
        parameters, another thread) is NOT seen by the packed-weight caches or the captured graph -- do not change weights inside sample().
        Holds nest (a depth counter): two samplers sharing one DiT instance each bracket their own call, the fingerprint is taken by the first
        and dropped by the last (the first one's `finally` used to pop the other's hold)."""
        d = self.__dict__
        depth = d.get("_pver_depth", 0)
        if active:
            if depth == 0 and os.environ.get("GVF_DIT_HOLD_PVER", "1") != "0":          # (=0: measurement switch)
                d["_pver_held"] = self._param_version()
            d["_pver_depth"] = depth + 1
        else:
            depth = max(0, depth - 1)
            d["_pver_depth"] = depth
            if depth == 0:
                d.pop("_pver_held", None)
        return self

    def _weights(self, lp=None):
        lp = self._lp() if lp is None else lp
        ver = (self._param_version(), lp)
        if self._wcache is not None and self._wcache["ver"] == ver:
            return self._wcache

        def prep(lin):
            w = lin.weight.detach().float().contiguous()
            return dit_ops.cast_pad(w, dit_ops.pad64(w.shape[1]), dtype=lp), \
                (None if lin.bias is None else lin.bias.detach().float().contiguous())

        def prep32(lin):
            return lin.weight.detach().float().contiguous(), (None if lin.bias is None else lin.bias.detach().float().contiguous())

        W = {"ver": ver, "lp": lp}
        # the small projections (0.3 % of the FLOPs, a third of the bf16 error: csrc/elem.hip) stay in fp32
        W["input_f32"], W["final_f32"] = prep32(self.input_layer), prep32(self.final_layer.linear)
        W["input_f32"] = (W["input_f32"][0].t().contiguous(), W["input_f32"][1])          # (Cin, C): the kernel copies it straight into LDS
        W["t0_f32"], W["t2_f32"] = prep32(self.t_embedder.mlp[0]), prep32(self.t_embedder.mlp[2])
        W["img_f32"], W["static_f32"] = prep32(self.image_cond_proj), prep32(self.static_cond_proj)
        W["input"] = prep(self.input_layer)
        W["t0"], W["t2"] = prep(self.t_embedder.mlp[0]), prep(self.t_embedder.mlp[2])
        W["img"], W["static"] = prep(self.image_cond_proj), prep(self.static_cond_proj)
        W["final"] = prep(self.final_layer.linear)
        # all adaLN projections of a step as one GEMM: rows = [blk0: 6C | 3C, blk1: ..., final: 2C]
        mods_w, mods_b, offs, o = [], [], [], 0
        for blk in self.blocks:
            parts = [blk.adaLN_modulation[-1]] + ([] if self.no_temporal_attn else [blk.adaLN_modulation_temporal[-1]])
            offs.append(o)
            for lin in parts:
                mods_w.append(lin.weight.detach().float())
                mods_b.append(lin.bias.detach().float())
                o += lin.out_features
        offs.append(o)
        mods_w.append(self.final_layer.adaLN_modulation[-1].weight.detach().float())
        mods_b.append(self.final_layer.adaLN_modulation[-1].bias.detach().float())
        W["mod_w_f32"] = torch.cat(mods_w).contiguous()
        W["mod_w"] = dit_ops.cast_pad(W["mod_w_f32"], dit_ops.pad64(self.model_channels), dtype=lp)
        W["mod_b"] = torch.cat(mods_b).contiguous()
        W["mod_offs"], W["mod_total"] = offs, W["mod_b"].numel()
        W["blocks"] = []
        for blk in self.blocks:
            b = {}
            for name in ("spatial_self_attn", "temporal_self_attn"):
                m = getattr(blk, name)
                if isinstance(m, MultiHeadAttention):
                    b[name] = dict(qkv=prep(m.to_qkv), out=prep(m.to_out), gq=m._gammas()[0], gk=m._gammas()[1])
                    b[name]["bounded"] = dit_ops.scores_bounded(*m._gammas(), head_dim=self.head_dim)     # fp16: no per-query shift needed (see attn_xt.hip)
            for name in ("image_cross_attn", "static_cross_attn"):
                m = getattr(blk, name)
                b[name] = dict(q=prep(m.to_q), kv=prep(m.to_kv), kv_f32=prep32(m.to_kv), out=prep(m.to_out), gq=m._gammas()[0], gk=m._gammas()[1])
                b[name]["bounded"] = dit_ops.scores_bounded(*m._gammas(), head_dim=self.head_dim)
            b["fc1"], b["fc2"] = prep(blk.mlp.mlp[0]), prep(blk.mlp.mlp[2])
            b["n3"] = (blk.norm3.weight.detach().float().contiguous(), blk.norm3.bias.detach().float().contiguous())
            b["n4"] = (blk.norm4.weight.detach().float().contiguous(), blk.norm4.bias.detach().float().contiguous())
            W["blocks"].append(b)
        self._wcache = W
        self._ctx_cache = {}
        return W

    def _rowblock_streams(self, W):
        """Packed weight streams of the row-block launches (csrc/rowblock.hip), built once per weight version: every launch reads ONE
        stream -- the projection that closes a sub-layer, [the MLP,] the projection that opens the next one -- in the order it consumes
        it.  Same 16-bit values as W's plain copies."""
        if "rb" in W:
            return W["rb"]
        P = dit_ops.rowblock_pack_stream
        blocks = W["blocks"]
        first = "spatial_self_attn"
        rb = {"in": P(None, w3=blocks[0][first]["qkv"][0]), "blocks": []}        # input_layer itself runs in fp32 (csrc/elem.hip)
        for i, b in enumerate(blocks):
            nxt = blocks[i + 1][first]["qkv"][0] if i + 1 < len(blocks) else None
            d = {}
            if self.no_temporal_attn:
                d["s2"] = P(b["spatial_self_attn"]["out"][0], w3=b["image_cross_attn"]["q"][0])
            else:
                d["s2"] = P(b["spatial_self_attn"]["out"][0], w3=b["temporal_self_attn"]["qkv"][0])
                d["s3"] = P(b["temporal_self_attn"]["out"][0], w3=b["image_cross_attn"]["q"][0])
                d["s23"] = P(b["spatial_self_attn"]["out"][0], temporal=(b["temporal_self_attn"]["qkv"][0], b["temporal_self_attn"]["out"][0]),
                             w3=b["image_cross_attn"]["q"][0])
            d["s4"] = P(b["image_cross_attn"]["out"][0], w3=b["static_cross_attn"]["q"][0])
            d["s5"] = P(b["static_cross_attn"]["out"][0], mlp=(b["fc1"][0], b["fc2"][0]), w3=nxt)
            rb["blocks"].append(d)
        W["rb"] = rb
        return rb

    # ---- step-invariant condition products ------------------------------------------------------------
    @staticmethod
    def _key(t):
        return None if t is None else (t.data_ptr(), t._version, tuple(t.shape), t.dtype)

    @staticmethod
    def _same_tensors(held, now):
        """True iff every tensor of `now` is the tensor (or a same-shape view of the storage) held in `held`,
        unmodified.  `held` are STRONG references taken when the cache entry was built: while they are alive the
        allocator cannot hand their addresses to another tensor, so (address, version, shape, dtype) identifies the
        contents -- a raw-pointer key without the references does not (a later sample's conditions land on the
        recycled addresses with version 0 and would silently hit)."""
        if held is None or len(held[0]) != len(now):
            return False
        refs, keys = held
        for r, k, t in zip(refs, keys, now):
            if (r is None) != (t is None):
                return False
            if t is not None and DiT._key(t) != k:        # same storage address (pinned by r), version, shape, dtype
                return False
        return True

    def invalidate_conditions(self):
        """Drop the step-invariant condition products (and the references that pin the condition tensors)."""
        self._ctx_cache = {}
        self._graph = None

    def prepare_conditions(self, cond_images, static_latent, deformation_position_xyz, T: int):
        """image_emb / static_emb -> per-block cross-attention K,V (tile images in the compute dtype), and the APE.  Cached on the identity of the
        three condition tensors; the entry HOLDS them (see _same_tensors), so a hit is a proof of equal contents.
        gvfdiffusion_amd's model_wrapper hands the same (concatenated) condition tensors to every step; a wrapper
        that rebuilds them per call still gets correct results, only without the reuse."""
        W = self._weights()
        lp = W["lp"]
        conds = (cond_images, static_latent, deformation_position_xyz)
        if self._ctx_cache.get("T") == T and self._ctx_cache.get("lp") == lp and self._same_tensors(self._ctx_cache.get("held"), conds):
            return self._ctx_cache
        C = self.model_channels
        dev = cond_images.device
        B, Tc, Li, Ci = cond_images.shape
        Ls = static_latent.shape[1]
        ctx = {"T": T, "lp": lp, "held": (conds, tuple(self._key(t) for t in conds)), "Li": Li, "Ls": Ls}
        # Round 6: the cache's device buffers (the 24 K / V tile images, the position embedding) PERSIST across condition sets of one shape: a new
        # sample refills them in place, so a captured hipGraph that reads them stays valid -- the next sample costs its condition products
        # (~3.1 ms at the released config), not those + an eager forward + a re-capture (~16 ms; measured 199.5 -> 215.8 ms per sample with
        # fresh condition tensors).  `epoch` names the buffer set; the graph is keyed on it (_forward_graphed).  Everything that reads the
        # buffers runs on the caller's stream, behind the refill.
        shape_key = (T, lp, str(dev), tuple(cond_images.shape), tuple(static_latent.shape),
                     None if deformation_position_xyz is None else tuple(deformation_position_xyz.shape), self.head_dim, self.pe_mode)
        old = self._ctx_cache if self._ctx_cache.get("shape_key") == shape_key else None
        ctx["shape_key"] = shape_key
        if old is not None:
            ctx["epoch"] = old["epoch"]
        else:
            ctx["epoch"] = next(_CTX_EPOCHS)               # (itertools.count: unique across instances and host threads)
        # Step-invariant: the condition projections and every block's to_kv(context) (model/dit.py:464-465, model/attention/modules.py:134-143)
        # at fp32-class accuracy on the bf16 matrix pipe -- both operands as two-term bf16 expansions laid out along K (dit_ops.split3_bf16:
        # a w^T = a_hi w_hi^T + a_lo w_hi^T + a_hi w_lo^T + O(2^-16)), ONE plain gvf_gemm with fp32 accumulation and fp32 output per projection
        # (three products at the 16-bit MFMA rate against one at v_mfma_f32's sixteenth of it; rounds 2-4 called rocBLAS here: ~5 ms per
        # sample).  Every output row is summed in one fixed order (no split-K), so a sample's numbers do not depend on what it is batched
        # with (tests: batch of three == three single samples, bit for bit).  Then ONE rounding to 16 bits when the cache builder folds the
        # softmax scale in and stores the tiled image the attention workgroups stage into LDS (csrc/attn_xt.hip); static K/V once per
        # sample, not per frame.
        H = self.num_heads
        S3 = self._split3_weights(W)
        f32 = torch.float32
        img_emb = torch.empty((B * Tc * Li, C), dtype=f32, device=dev)
        st_emb = torch.empty((B * Ls, C), dtype=f32, device=dev)
        dit_ops.gemm(dit_ops.split3_bf16(cond_images.reshape(B * Tc * Li, Ci).float().contiguous()), S3["img"], W["img_f32"][1], img_emb, dit_ops.EPI_STORE_F32)
        dit_ops.gemm(dit_ops.split3_bf16(static_latent.reshape(B * Ls, -1).float().contiguous()), S3["static"], W["static_f32"][1], st_emb, dit_ops.EPI_STORE_F32)
        # fp16 caches keep the 64 largest-norm keys of every (set, head) in the first tile (gvf_attn_key_order / gvf_attn_pack_kv_ordered): the kernel's per-query shift is the best
        # score against the FIRST key tile and a later key that beats it by 2^16 costs the workgroup an exact pass -- on trained-like scores
        # 38 % of the workgroups with the keys in context order, ~0 with the high-norm keys (attention sinks, artefact tokens) in front
        # (tests/test_dit_fp16_gpu.py::test_full_config_trained_like_weights).  bf16 needs no shift: context order (GVF_DIT_KEY_ORDER=0/1 forces).
        want = os.environ.get("GVF_DIT_KEY_ORDER")
        ordered = (lp == torch.float16) if want is None else want == "1"
        for name, emb, n_sets, L in (("kv_img", img_emb, B * Tc, Li), ("kv_st", st_emb, B, Ls)):
            ctx[name] = self._context_kv(ctx, old, name, dit_ops.split3_bf16(emb), S3[name + "_all"], W[name + "_bias_all"], W[name + "_gk_all"],
                                         n_sets, L, ordered and L <= 8192)
        if self.pe_mode == "ape":
            assert deformation_position_xyz is not None, "Deformation position xyz is required for APE mode"
            pos = self.pos_embedder(deformation_position_xyz).float().contiguous()      # (B, N, C)
        elif self.pe_mode == "learnable":
            pos = self.pos_embedder.detach().float().expand(B, -1, -1).contiguous()
        else:
            pos = None
        if old is not None and pos is not None and old.get("pos") is not None and old["pos"].shape == pos.shape:
            old["pos"].copy_(pos)
            pos = old["pos"]
        elif old is not None and (pos is None) != (old.get("pos") is None):
            ctx["epoch"] = next(_CTX_EPOCHS)               # (cannot happen for one shape_key; never leave a graph pointing at a dropped buffer)
        ctx["pos"] = pos
        self._ctx_cache = ctx
        return ctx

    def _context_kv(self, ctx, old, name, x3, w_all, bias_all, gk_all, n_sets, L, ordered):
        """Every block's to_kv(context) (model/attention/modules.py:134-143 inside model/dit.py:257-262) for ONE context, all blocks at once: the
        projections as one wide GEMM per chunk of blocks (weights stacked along N; x3 = the context's [hi | lo | hi] rows, shared), then ONE key
        order launch and ONE cache-builder launch per chunk (gvf_attn_*_groups: a block = a column band of the wide product).  Round 6: block by
        block these were 12 + 12 + 12 launches per context, the order kernel latency-bound at 65 us a launch (1.55 ms of the 4.7 ms a new sample's
        conditions cost).  A chunk's fp32 product is bounded (~2 GiB; the released shapes at B = 1: image context 1.6 GB = all 12 blocks).
        Returns the per-block list the forward reads: (k_tiles, v_tiles) views of one persistent buffer pair, or row-major 16-bit kv (head_dim 64)."""
        C, H, lp = self.model_channels, self.num_heads, ctx["lp"]
        nb = len(self.blocks)
        M = x3.shape[0]
        per_block = M * 2 * C * 4
        G = max(1, min(nb, int(os.environ.get("GVF_DIT_KV_CHUNK_BYTES", 2 << 30)) // max(per_block, 1)))       # (the variable: a measurement switch)
        # the kernel is chosen from N and K alone, so that a sample's numbers do not depend on what it is batched with
        g8 = lambda n_: os.environ.get("GVF_GEMM8", "1") != "0" and dit_ops.gemm8_eligible(256, n_, x3.shape[1], x3.stride(0), w_all.stride(0), n_, dit_ops.EPI_STORE_F32) != 0
        tiled = self.head_dim == 32
        if tiled:
            nbytes = n_sets * H * ((L + 63) // 64) * 4096
            bufs = old.get(name + "_all") if old is not None else None
            if bufs is None:
                bufs = (torch.empty((nb, nbytes), dtype=torch.uint8, device=x3.device), torch.empty((nb, nbytes), dtype=torch.uint8, device=x3.device))
            ctx[name + "_all"] = bufs
        out = []
        for g0 in range(0, nb, G):
            g1 = min(nb, g0 + G)
            n = (g1 - g0) * 2 * C
            wide = torch.empty((M, n), dtype=torch.float32, device=x3.device)
            (dit_ops.gemm8 if g8(n) else dit_ops.gemm)(x3, w_all[g0 * 2 * C:g1 * 2 * C], None if bias_all is None else bias_all[g0 * 2 * C:g1 * 2 * C], wide,
                                                       dit_ops.EPI_STORE_F32)
            if not tiled:
                # head_dim 64: the strided flash attention reads row-major [k | v] rows of the operand type (one rounding of the fp32-class
                # projection, as the tile image's); MultiHeadRMSNorm of k is the kernel's prologue (gamma_k)
                for j in range(g1 - g0):
                    band = wide[:, j * 2 * C:(j + 1) * 2 * C]
                    if old is not None:
                        old[name][g0 + j].copy_(band)
                        out.append(old[name][g0 + j])
                    else:
                        out.append(band.to(lp).contiguous())
                continue
            bands = wide.view(M, g1 - g0, 2 * C).permute(1, 0, 2)            # (groups, rows, 2C): group stride 2C, row stride n
            order = dit_ops.key_order_by_norm_groups(bands, g1 - g0, n_sets, L, H, 0) if ordered else None
            dit_ops.attention_pack_kv_groups(bands, g1 - g0, n_sets, L, H, 0, C, gamma_k=None if gk_all is None else gk_all[g0:g1], dtype=lp,
                                             key_order=order, out=(bufs[0][g0:g1], bufs[1][g0:g1]))
            out.extend((bufs[0][g], bufs[1][g]) for g in range(g0, g1))
        return out

    def _split3_weights(self, W):
        """[hi | hi | lo] bf16 expansions of the fp32 weights of the hoisted projections (dit_ops.split3_bf16), once per weight version."""
        S3 = W.get("split3")
        if S3 is None:
            # the 12 blocks' to_kv weights of a context stacked along N: one wide projection per context (_context_kv)
            stack = lambda name: dit_ops.split3_bf16(torch.cat([b[name]["kv_f32"][0] for b in W["blocks"]], 0).contiguous(), weights=True)
            S3 = W["split3"] = {"img": dit_ops.split3_bf16(W["img_f32"][0], weights=True), "static": dit_ops.split3_bf16(W["static_f32"][0], weights=True),
                                "kv_img_all": stack("image_cross_attn"), "kv_st_all": stack("static_cross_attn")}
            for key, name in (("kv_img", "image_cross_attn"), ("kv_st", "static_cross_attn")):
                biases = [b[name]["kv_f32"][1] for b in W["blocks"]]
                W[key + "_bias_all"] = None if biases[0] is None else torch.cat(biases, 0).contiguous()
                gks = [b[name]["gk"] for b in W["blocks"]]
                W[key + "_gk_all"] = None if gks[0] is None else torch.stack([g.reshape(-1) for g in gks], 0).float().contiguous()
            # the fp32 matrices were only the source of the expansions: drop them, keep the biases (26 fp32 matrices stayed resident
            # beside their [hi | hi | lo] copies in every DiT instance in flight).  The parameters themselves are untouched.
            W["img_f32"], W["static_f32"] = (None, W["img_f32"][1]), (None, W["static_f32"][1])
            for b in W["blocks"]:
                for name in ("image_cross_attn", "static_cross_attn"):
                    b[name]["kv_f32"] = (None, b[name]["kv_f32"][1])
        return S3

    # ---- what depends on the timestep alone ---------------------------------------------------------------
    @torch.no_grad()
    def precompute_modulation(self, t: torch.Tensor):
        """Timestep embedding + every adaLN projection (model/dit.py:449-453 and the 12 x adaLN_modulation of :236-238) for ALL the model-input
        times in `t` (K,) in ONE pair of launches: a table (K, mod_total) fp32.  A sampler with a fixed time grid calls this before its first
        evaluation (DPM_Solver.sample via model_wrapper's prepare_times); a later forward whose time tensor carries its host values
        (`gvf_host_values`, set by model_wrapper) and finds them in the table skips the two launches at the head of the step -- the 115 MB of
        fp32 projection weights cross HBM once per sample instead of once per step.  Same kernels, same numbers; any other call computes the
        modulation inside the forward as before."""
        dev = next(self.parameters()).device
        if dev.type != "cuda" or not self.modulation_table:
            return
        W = self._weights()
        fdim, C = self.t_embedder.frequency_embedding_size, self.model_channels
        if not (fdim % 4 == 0 and fdim <= 1024 and C % 4 == 0 and C <= 1024):
            return
        th = t.detach().reshape(-1).float().cpu()
        tab = getattr(self, "_mod_table", None)
        if tab is not None and tab["version"] == (self._param_version(), self._lp()) and all(float(v) in tab["rows"] for v in th.tolist()):
            return                                     # the same grid as the last sample's: the table is a per-JOB cost (~12 ms for 33 rows), not per sample
        s2 = dit_ops.timestep_embed_f32(th.to(dev).contiguous(), *W["t0_f32"], *W["t2_f32"], freq_dim=fdim)
        mod = dit_ops.modulation_f32(s2, W["mod_w_f32"], W["mod_b"])
        self._mod_table = {"version": (self._param_version(), self._lp()), "rows": {float(v): i for i, v in enumerate(th.tolist())}, "mod": mod,
                           "arange": torch.arange(mod.shape[0], device=dev)}

    def _mod_from_table(self, t, B, pv=None):
        """Row numbers (B,) int64 on the device of this forward's times in the precomputed table, or None (no table / unknown time / no host
        values).  A view of the table's arange for the usual B = 1 and guided B = 3 (equal times) calls: no launch, no host -> device copy."""
        tab, hv = getattr(self, "_mod_table", None), getattr(t, "gvf_host_values", None)
        if tab is None or hv is None or len(hv) != B or tab["version"] != (self._param_version() if pv is None else pv, self._lp()):
            return None
        try:
            idx = [tab["rows"][v] for v in hv]
        except KeyError:
            return None
        self.__dict__["mod_table_hits"] = self.__dict__.get("mod_table_hits", 0) + 1      # (read by tests: does a caller's solver reach the table?)
        if all(i == idx[0] for i in idx):
            return tab["arange"][idx[0]:idx[0] + 1].expand(B)
        return tab["arange"][torch.tensor(idx)]

    def count_attention_fallbacks(self, on: bool = True):
        """Instrument the tiled attention launches (spatial self, image cross, static cross: 3 per block) with the C ABI's `fallback_counter`
        (include/gvf_dit.h): += 1 per 256-query workgroup whose max-free softmax tripped its range guard and was recomputed on the exact
        running-maximum path.  The counter's address is part of a captured graph, so switching drops the graph.  Read with
        attention_fallbacks(); attention_workgroups(B, T, N) is the number of workgroups one forward launches."""
        if on:
            dev = next(self.parameters()).device
            self._fallbacks = torch.zeros(1, dtype=torch.int32, device=dev)
        else:
            self._fallbacks = None
        self._graph = None
        return self

    def attention_fallbacks(self, reset: bool = True) -> int:
        """Workgroups that took the exact path since the last reset (one device -> host read)."""
        if self._fallbacks is None:
            raise RuntimeError("count_attention_fallbacks() first")
        n = int(self._fallbacks.item())
        if reset:
            self._fallbacks.zero_()
        return n

    def attention_workgroups(self, B: int, T: int, N: int) -> int:
        return len(self.blocks) * 3 * B * T * self.num_heads * ((N + 255) // 256)

    # ---- forward ------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, t: torch.Tensor, cond_images: torch.Tensor, static_latent: torch.Tensor,
                deformation_position_xyz: torch.Tensor = None) -> torch.Tensor:
        if self.use_graph:
            return self._forward_graphed(x, t, cond_images, static_latent, deformation_position_xyz)
        return self._forward(x, t, cond_images, static_latent, deformation_position_xyz)

    def enable_graph(self, on: bool = True):
        """Capture the ~200 launches of one forward pass into a hipGraph on first use and replay it for
        every later step with the same shapes and the same condition tensors (the sampling loop's case):
        removes the per-launch host cost from the denoise step.  Numerics are unchanged (same kernels)."""
        self.use_graph = bool(on)
        self._graph = None
        return self

    @torch.no_grad()
    def _forward_graphed(self, x, t, cond_images, static_latent, deformation_position_xyz=None):
        _lib.require_cuda(x, t, cond_images, static_latent, deformation_position_xyz)
        pv = self._param_version()                           # once per forward: 0.1-0.4 ms of host time
        mod_rows = self._mod_from_table(t, x.shape[0], pv)   # (looked up from the HOST values the tensor carries: no read-back)
        t = t.to(x.device)
        # the step-invariant condition products: a hit for every step of a sample (identity of the three tensors); a NEW sample's are computed
        # here, eagerly, INTO the buffers the captured graph reads (prepare_conditions keeps them per shape: `epoch`), so the graph is replayed
        # for sample after sample and re-captured only when the buffer set itself changes (another shape, operand type or weight version)
        ctx = self.prepare_conditions(cond_images, static_latent, deformation_position_xyz, x.shape[1])
        key = (tuple(x.shape), x.dtype, tuple(t.shape), t.dtype, pv, self._lp(),
               None if mod_rows is None else self._mod_table["mod"].data_ptr(), ctx["epoch"])
        conds = (cond_images, static_latent, deformation_position_xyz)
        g = self._graph
        if g is None or g["key"] != key:
            sx, st = x.clone(), t.clone()
            smod = None if mod_rows is None else mod_rows.clone()      # the row numbers: the graph gathers its rows of the table itself
            # One capture at a time per process, and in thread-local capture mode, so that a capture on one host thread does not fail because
            # another thread (another sample in flight on its own stream and DiT instance, utils/in_flight.py) launches or allocates meanwhile.
            # (Captures in flight are bit-identical to serial sampling: tests/test_inference_script_gpu.py, scripts/inflight_capture_repro.py.)
            with _CAPTURE_LOCK:
                # eager run first: builds the weight / condition caches and warms the allocator outside the capture
                self._forward(sx, st, cond_images, static_latent, deformation_position_xyz, mod_idx=smod)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    sy = self._forward(sx, st, cond_images, static_latent, deformation_position_xyz, mod_idx=smod)
            g = self._graph = {"key": key, "held": (conds, tuple(self._key(c) for c in conds)), "graph": graph, "x": sx,
                               "t": st, "y": sy, "mod": smod}
        g["x"].copy_(x)
        if g["mod"] is not None:
            g["mod"].copy_(mod_rows)                           # 8 bytes per sample, like the time itself; the graph gathers the rows (225 KB) of the table
        else:
            g["t"].copy_(t)
        g["graph"].replay()
        return g["y"].clone()

    def _forward_with_mem_ratio(self, x, t, cond_images, static_latent, deformation_position_xyz=None, mem_ratio=1.0):
        """ElasticModule contract at inference (elastic_utils.py:166-168): (exact_mem_ratio, output)."""
        return 1.0, self._forward(x, t, cond_images, static_latent, deformation_position_xyz)

    @torch.no_grad()
    def _forward(self, x, t, cond_images, static_latent, deformation_position_xyz=None, mod_idx=None):
        """mod_idx: optional (B,) row numbers into precompute_modulation's table (the graphed forward hands its static copy in)."""
        _lib.require_cuda(x, t, cond_images, static_latent, deformation_position_xyz)
        B, T, N, Cin = x.shape
        C, H = self.model_channels, self.num_heads
        dev = x.device
        M = B * T * N
        W = self._weights()
        ctx = self.prepare_conditions(cond_images, static_latent, deformation_position_xyz, T)
        Li, Ls = ctx["Li"], ctx["Ls"]
        bf, f32 = W["lp"], torch.float32              # bf: the 16-bit operand type of this forward (bf16 or fp16)

        # timestep embedder (sinusoid, two Linears, two SiLUs: one launch) and every adaLN projection of the step (one GEMV), both in fp32
        fdim = self.t_embedder.frequency_embedding_size
        if mod_idx is None and not self.use_graph:
            mod_idx = self._mod_from_table(t, B)           # eager calls look the step up here (the graphed forward hands its copy in)
        if mod_idx is not None:
            mod = self._mod_table["mod"].index_select(0, mod_idx)
            assert mod.shape == (B, W["mod_total"]) and mod.dtype == f32
        elif fdim % 4 == 0 and fdim <= 1024 and C % 4 == 0 and C <= 1024:
            s2 = dit_ops.timestep_embed_f32(t.to(dev).float().contiguous(), *W["t0_f32"], *W["t2_f32"], freq_dim=fdim)
            mod = dit_ops.modulation_f32(s2, W["mod_w_f32"], W["mod_b"])
        else:
            mod = torch.empty((B, W["mod_total"]), dtype=f32, device=dev)
            tf = dit_ops.cast_pad(TimestepEmbedder.timestep_embedding(t.to(dev), fdim).contiguous(), dit_ops.pad64(fdim), dtype=bf)
            h1 = torch.empty((B, C), dtype=f32, device=dev)
            dit_ops.gemm(tf, *W["t0"], h1, dit_ops.EPI_STORE_F32)
            t_emb = torch.empty((B, C), dtype=f32, device=dev)
            dit_ops.gemm(dit_ops.cast_pad(h1, dit_ops.pad64(C), act=1, dtype=bf), *W["t2"], t_emb, dit_ops.EPI_STORE_F32)
            dit_ops.gemm(dit_ops.cast_pad(t_emb, dit_ops.pad64(C), act=1, dtype=bf), W["mod_w"], W["mod_b"], mod, dit_ops.EPI_STORE_F32)
        mod_ld = W["mod_total"]

        # residual stream h (fp32) = position embedding broadcast over T + input_layer(x), in fp32 (csrc/elem.hip)
        small_f32 = C <= 512 and C % 4 == 0 and Cin <= 24 and self.out_channels <= 32
        x2d = x.reshape(M, Cin).float().contiguous()
        pos = None if ctx["pos"] is None else ctx["pos"].reshape(B * N, C)
        # the row-block launches work on whole 48-row blocks of one sample: a sample whose T * N is not a multiple of 48 (T = 16, 32 at
        # N = 512) gets its rows padded (needs whole 64-key tiles per frame for the attention's strided views: N % 64 == 0)
        TNp = dit_ops.rowblock_padded_rows(T * N)
        if self.use_rowblock and self.head_dim == 32 and small_f32 and Cin % 4 == 0 and Cin <= 16 and dit_ops.rowblock_supported(C, TNp, int(C * self.mlp_ratio)) \
                and (TNp == T * N or (N % 64 == 0 and self.rowblock_tiled_kv)):
            y = self._blocks_rowblock(x2d, mod, mod_ld, W, ctx, B, T, N, pos)          # its first launch also does input_layer
            return y.to(x.dtype if x.dtype.is_floating_point else f32)
        h = torch.empty((M, C), dtype=f32, device=dev)
        if small_f32:
            dit_ops.input_layer_f32(x2d, W["input_f32"][0], W["input_f32"][1], h, pos=pos, pos_period=N, rows_per_group=T * N)
        elif ctx["pos"] is not None:
            h.copy_(ctx["pos"][:, None].expand(B, T, N, C).reshape(M, C))
        else:
            h.zero_()
        xb = None if small_f32 else dit_ops.cast_pad(x.reshape(M, Cin).float().contiguous(), dit_ops.pad64(Cin), dtype=bf)
        hb = torch.empty((M, C), dtype=bf, device=dev)          # attention-output scratch of the cross attentions
        qkv = torch.empty((M, 3 * C), dtype=bf, device=dev)
        ab = torch.empty((M, C), dtype=bf, device=dev)          # attention output / q projection
        hidden = torch.empty((M, int(C * self.mlp_ratio)), dtype=bf, device=dev)
        TN = T * N
        hd = self.head_dim
        nb_self = B * T * H * ((N + 63) // 64) * 4096
        kv_self = (torch.empty(nb_self, dtype=torch.uint8, device=dev), torch.empty(nb_self, dtype=torch.uint8, device=dev))
        # LayerNorm is folded into the GEMMs around it (csrc/gemm.hip): every update of the stream x = x + g * h also writes
        # the row statistics of the new x, and the projection that follows normalises its A operand on the fly
        fuse_ln = self.fuse_layernorm != 0 and TN % 128 == 0 and C % 128 == 0 and C <= 1024
        fuse_max_n = 1 << 30 if self.fuse_layernorm == 1 else 512
        n_part = dit_ops.gemm_stats_parts(C)
        stats = torch.empty((M, n_part, 2), dtype=f32, device=dev) if fuse_ln else None
        have_stats = [False]       # the row statistics exist once a residual GEMM has written them: the fp32 input layer (csrc/elem.hip) does
                                   # not, so block 0's first projection normalises with the LayerNorm launch

        def resid(a_, wb, gate=None):
            """h += gate * (a_ @ W^T + b)  (+ statistics of the new h)"""
            kw = dict(gate=gate, gate_ld=mod_ld, rows_per_group=TN) if gate is not None else {}
            if fuse_ln:
                dit_ops.gemm_resid_stats(a_, wb[0], wb[1], h, stats, **kw)
                have_stats[0] = True
            else:
                dit_ops.gemm(a_, wb[0], wb[1], h, dit_ops.EPI_RESID_F32, **kw)

        def ln_gemm(wb, out, epi, ln_w=None, ln_b=None, shift=None, scale=None):
            """out = epi((LN(h) * s + t) @ W^T + b)"""
            if fuse_ln and have_stats[0] and wb[0].shape[0] <= fuse_max_n:
                dit_ops.gemm_ln_bf16(h, stats, n_part, wb[0], wb[1], out, epi, 1e-6, ln_w, ln_b, shift, scale, mod_ld, TN)
            else:
                dit_ops.layernorm_modulate(h, hb, 1e-6, ln_w, ln_b, shift, scale, mod_ld, TN)
                dit_ops.gemm(hb, wb[0], wb[1], out, epi)

        if not small_f32:
            resid(xb, W["input"])           # h = pos + input_layer(x)

        def mview(off):                                       # (B,) rows of `mod`, columns [off, off+C)
            return mod[:, off:]

        for i, b in enumerate(W["blocks"]):
            o = W["mod_offs"][i]
            sh_s, sc_s, g_s, sh_m, sc_m, g_m = (mview(o + k * C) for k in range(6))
            # -- spatial self attention over N
            a = b["spatial_self_attn"]
            ln_gemm(a["qkv"], qkv, dit_ops.EPI_STORE_16, shift=sh_s, scale=sc_s)
            if hd != 32:
                s3 = (N * 3 * C, 0, 3 * C)
                dit_ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], ab, B * T, 1, N, N, H, s3, s3, s3, (N * C, 0, C), a["gq"], a["gk"], head_dim=hd)
            else:
                # K (RMS-normed, pre-scaled) and V^T of this step's projection into the tiled image, then the tiled-cache kernel
                dit_ops.attention_pack_kv(qkv, B * T, N, H, C, 2 * C, gamma_k=a["gk"], out=kv_self)
                dit_ops.attention_tiled(qkv, kv_self[0], kv_self[1], ab, B * T, 1, N, N, H, (N * 3 * C, 0, 3 * C), (N * C, 0, C), 1, 0,
                                        gamma_q=a["gq"], bounded=a["bounded"], fallback_counter=self._fallbacks)
            resid(ab, a["out"], g_s)
            # -- temporal self attention over T (strided views, no transposes)
            if not self.no_temporal_attn:
                sh_t, sc_t, g_t = (mview(o + (6 + k) * C) for k in range(3))
                a = b["temporal_self_attn"]
                ln_gemm(a["qkv"], qkv, dit_ops.EPI_STORE_16, shift=sh_t, scale=sc_t)
                st = (TN * 3 * C, 3 * C, N * 3 * C)           # outer = sample, inner = token, seq = frame
                dit_ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], ab, B, N, T, T, H, st, st, st, (TN * C, C, N * C), a["gq"], a["gk"], head_dim=hd)
                resid(ab, a["out"], g_t)
            # -- image cross attention (affine LayerNorm, cached K/V)
            a = b["image_cross_attn"]
            ln_gemm(a["q"], ab, dit_ops.EPI_STORE_16, ln_w=b["n3"][0], ln_b=b["n3"][1])
            if hd != 32:
                kv = ctx["kv_img"][i]                             # (B T Li, 2C) rows [k | v], one key set per (sample, frame)
                sk = (Li * 2 * C, 0, 2 * C)
                dit_ops.attention(ab, kv, kv[:, C:], hb, B * T, 1, N, Li, H, (N * C, 0, C), sk, sk, (N * C, 0, C), a["gq"], a["gk"], head_dim=hd)
            else:
                kt, vt = ctx["kv_img"][i]
                dit_ops.attention_tiled(ab, kt, vt, hb, B * T, 1, N, Li, H, (N * C, 0, C), (N * C, 0, C), 1, 0, gamma_q=a["gq"], bounded=a["bounded"], fallback_counter=self._fallbacks)
            resid(hb, a["out"])
            # -- static cross attention: K/V shared by the T frames of a sample (inner stride 0)
            a = b["static_cross_attn"]
            ln_gemm(a["q"], ab, dit_ops.EPI_STORE_16, ln_w=b["n4"][0], ln_b=b["n4"][1])
            if hd != 32:
                kv = ctx["kv_st"][i]                              # (B Ls, 2C): one key set per sample, shared by its T frames (inner stride 0)
                sk = (Ls * 2 * C, 0, 2 * C)
                dit_ops.attention(ab, kv, kv[:, C:], hb, B, T, N, Ls, H, (TN * C, N * C, C), sk, sk, (TN * C, N * C, C), a["gq"], a["gk"], head_dim=hd)
            else:
                kt, vt = ctx["kv_st"][i]
                dit_ops.attention_tiled(ab, kt, vt, hb, B, T, N, Ls, H, (TN * C, N * C, C), (TN * C, N * C, C), 1, 0, gamma_q=a["gq"], bounded=a["bounded"], fallback_counter=self._fallbacks)
            resid(hb, a["out"])
            # -- MLP
            ln_gemm(b["fc1"], hidden, dit_ops.EPI_GELU_16, shift=sh_m, scale=sc_m)
            resid(hidden, b["fc2"], g_m)

        o = W["mod_offs"][-1]
        y = torch.empty((M, self.out_channels), dtype=f32, device=dev)
        if small_f32:
            dit_ops.final_layer_f32(h, W["final_f32"][0], W["final_f32"][1], y, shift=mview(o), scale=mview(o + C), mod_ld=mod_ld, rows_per_group=TN)
        else:
            ln_gemm(W["final"], y, dit_ops.EPI_STORE_F32, shift=mview(o), scale=mview(o + C))
        return y.reshape(B, T, N, self.out_channels).to(x.dtype if x.dtype.is_floating_point else f32)

    def _blocks_rowblock(self, x2d, mod, mod_ld, W, ctx, B, T, N, pos):
        """The blocks with one launch per sub-layer boundary (csrc/rowblock.hip): per block  spatial attention | to_out + adaLN +
        to_qkv | temporal attention | to_out + norm3 + to_q | image attention | to_out + norm4 + to_q | static attention | to_out +
        adaLN + MLP + adaLN + the NEXT block's to_qkv (q row-major, K / V^T as the attention's tile images) -- 8 launches instead of 20, the fp32 stream through HBM 5 times instead of 15,
        the normalised rows and the MLP's hidden units never.  Same rounding points as the unfused path (and as oracle/dit_ref.py).
        Row layout: sample b owns rows [b TNp, b TNp + T N), TNp = T N rounded up to whole 48-row blocks; the padding rows (T N not a
        multiple of 48) are computed like any other row by the row-block launches and never read by an attention."""
        C, H = self.model_channels, self.num_heads
        TN = T * N
        TNp = dit_ops.rowblock_padded_rows(TN)
        padded = TNp != TN
        M = B * TNp
        dev = x2d.device
        bf, f32 = W["lp"], torch.float32              # bf: the 16-bit operand type (bf16 or fp16); the packed streams are of that type
        rb = self._rowblock_streams(W)
        Li, Ls = ctx["Li"], ctx["Ls"]
        # buffers an attention writes and a row-block launch reads whole: their padding rows must stay finite
        alloc = torch.zeros if padded else torch.empty
        h = torch.empty((M, C), dtype=f32, device=dev)            # the residual stream
        qkv = torch.empty((M, 3 * C), dtype=bf, device=dev)
        ab = alloc((M, C), dtype=bf, device=dev)                  # self-attention output
        hb = alloc((M, C), dtype=bf, device=dev)                  # cross-attention output
        qb = torch.empty((M, C), dtype=bf, device=dev)            # q projection of the cross attentions
        if padded:
            xp = torch.zeros((B, TNp, x2d.shape[1]), dtype=f32, device=dev)
            xp[:, :TN] = x2d.view(B, TN, -1)
            x2d = xp.view(M, -1)
        nb_self = B * T * H * ((N + 63) // 64) * 4096
        kv_self = (torch.empty(nb_self, dtype=torch.uint8, device=dev), torch.empty(nb_self, dtype=torch.uint8, device=dev))
        hidden_units = int(C * self.mlp_ratio)
        blocks = W["blocks"]
        offs = W["mod_offs"]

        def mview(off):
            return mod[:, off:]

        def fused(a_, stream, **kw):
            dit_ops.rowblock_fused(a_, stream, h, mod_ld=mod_ld, rows_per_group=TNp, eps=1e-6, dtype=bf, **kw)

        # every attention launch also touches the packed weights of the row-block launch behind it (GVF_DIT_PREFETCH=0: off): the 107 MB of
        # weights of a forward cycle through the caches once per step, so each launch starts on cold weights -- while the attention before it,
        # bound by its matrix / vector pipes, leaves the memory system idle
        pf = (lambda t: t) if self.weight_prefetch else (lambda t: None)

        o = offs[0]
        # h = pos + input_layer(x); adaLN of block 0; its to_qkv
        # to_qkv of the spatial self attention: with whole 64-key tiles per frame its launch writes q row-major and K / V^T directly as the
        # tiled images the attention kernel stages (no row-major k, v; no pack launch)
        tiled_kv = N % 64 == 0 and self.rowblock_tiled_kv
        R = dit_ops.ROWBLOCK_ROWS
        # (a padded sample's padding rows are exactly the phantom tokens of its last block: ceil(N / (R / T)) * R == TNp)
        temporal_fused = self.rowblock_temporal and not self.no_temporal_attn and R % T == 0
        qs = torch.empty((M, C), dtype=bf, device=dev) if tiled_kv else None

        def qkv_out(blk):
            a_ = blk["spatial_self_attn"]
            if tiled_kv:
                return dict(out3=qs, b3=a_["qkv"][1], kv_tiles=kv_self, kv_L=N, gamma_k=a_["gk"], kv_group_rows=TN if padded else 0)
            return dict(out3=qkv, b3=a_["qkv"][1])

        # per-frame views of the row buffers.  Padded: outer = sample (stride TNp rows), inner = frame (N rows), key set of (b, f) = b T + f;
        # otherwise the frames of all samples are one uniform run (the launch geometry the kernels were tuned on)
        if padded:
            fr = dict(n=(B, T), c=(TNp * C, N * C, C), c3=(TNp * 3 * C, N * 3 * C, 3 * C), kv=(T, 1))
        else:
            fr = dict(n=(B * T, 1), c=(N * C, 0, C), c3=(N * 3 * C, 0, 3 * C), kv=(1, 0))
        fr_c = (TNp * C, N * C, C)                                 # (sample, frame) view for the per-sample key set of the static attention

        # h = pos (broadcast over the frames) + input_layer(x) in fp32, adaLN of block 0 and its to_qkv: one launch
        if pos is None:
            h.zero_()
        fused(None, rb["in"], ln1=dict(shift=mview(o), scale=mview(o + C)), in_x=x2d, in_wt=W["input_f32"][0], in_b=W["input_f32"][1],
              x_in=pos, x_in_period=N, **qkv_out(blocks[0]))
        for i, b in enumerate(blocks):
            o, s = offs[i], rb["blocks"][i]
            g_s, sh_m, sc_m, g_m = mview(o + 2 * C), mview(o + 3 * C), mview(o + 4 * C), mview(o + 5 * C)
            n3, n4 = dict(ln_w=b["n3"][0], ln_b=b["n3"][1]), dict(ln_w=b["n4"][0], ln_b=b["n4"][1])
            a = b["spatial_self_attn"]
            if tiled_kv:
                dit_ops.attention_tiled(qs, kv_self[0], kv_self[1], ab, *fr["n"], N, N, H, fr["c"], fr["c"], *fr["kv"], gamma_q=a["gq"], bounded=a["bounded"], fallback_counter=self._fallbacks,
                                        prefetch=pf(s["s23"] if (temporal_fused and not self.no_temporal_attn) else s["s2"]))
            else:                                              # (never padded: see _forward)
                dit_ops.attention_pack_kv(qkv, B * T, N, H, C, 2 * C, gamma_k=a["gk"], out=kv_self)
                dit_ops.attention_tiled(qkv, kv_self[0], kv_self[1], ab, *fr["n"], N, N, H, fr["c3"], fr["c"], *fr["kv"], gamma_q=a["gq"], bounded=a["bounded"], fallback_counter=self._fallbacks)
            ai = b["image_cross_attn"]
            if self.no_temporal_attn:
                fused(ab, s["s2"], b1=a["out"][1], gate1=g_s, ln1=n3, out3=qb, b3=ai["q"][1])
            else:
                sh_t, sc_t, g_t = (mview(o + (6 + k) * C) for k in range(3))
                at = b["temporal_self_attn"]
                if temporal_fused:
                    fused(ab, s["s23"], b1=a["out"][1], gate1=g_s, ln1=dict(shift=sh_t, scale=sc_t), out3=qb, b3=ai["q"][1],
                          temporal=dict(frames=T, stride=N, b_qkv=at["qkv"][1], gamma_q=at["gq"], gamma_k=at["gk"], b_out=at["out"][1], gate=g_t, ln=n3))
                else:
                    fused(ab, s["s2"], b1=a["out"][1], gate1=g_s, ln1=dict(shift=sh_t, scale=sc_t), out3=qkv, b3=at["qkv"][1])
                    st = (TNp * 3 * C, 3 * C, N * 3 * C)          # outer = sample, inner = token, seq = frame
                    dit_ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], ab, B, N, T, T, H, st, st, st, (TNp * C, C, N * C), at["gq"], at["gk"])
                    fused(ab, s["s3"], b1=at["out"][1], gate1=g_t, ln1=n3, out3=qb, b3=ai["q"][1])
            kt, vt = ctx["kv_img"][i]
            dit_ops.attention_tiled(qb, kt, vt, hb, *fr["n"], N, Li, H, fr["c"], fr["c"], *fr["kv"], gamma_q=ai["gq"], bounded=ai["bounded"], fallback_counter=self._fallbacks, prefetch=pf(s["s4"]))
            ast = b["static_cross_attn"]
            fused(hb, s["s4"], b1=ai["out"][1], ln1=n4, out3=qb, b3=ast["q"][1])
            kt, vt = ctx["kv_st"][i]
            dit_ops.attention_tiled(qb, kt, vt, hb, B, T, N, Ls, H, fr_c, fr_c, 1, 0, gamma_q=ast["gq"], bounded=ast["bounded"], fallback_counter=self._fallbacks, prefetch=pf(s["s5"]))
            kw = dict(b1=ast["out"][1], ln1=dict(shift=sh_m, scale=sc_m), mlp_bias=(b["fc1"][1], b["fc2"][1]), hidden=hidden_units, gate_m=g_m)
            if i + 1 < len(blocks):
                on = offs[i + 1]
                fused(hb, s["s5"], ln2=dict(shift=mview(on), scale=mview(on + C)), **qkv_out(blocks[i + 1]), **kw)
            else:
                fused(hb, s["s5"], **kw)                       # the last MLP; final_layer reads the stream itself
            if self.capture_blocks is not None:
                self.capture_blocks.append(h.view(B, TNp, C)[:, :TN].reshape(B, T, N, C).clone())
        on = offs[-1]
        y = torch.empty((M, self.out_channels), dtype=f32, device=dev)
        dit_ops.final_layer_f32(h, W["final_f32"][0], W["final_f32"][1], y, shift=mview(on), scale=mview(on + C), mod_ld=mod_ld, rows_per_group=TNp)
        return y.view(B, TNp, self.out_channels)[:, :TN].reshape(B, T, N, self.out_channels)
