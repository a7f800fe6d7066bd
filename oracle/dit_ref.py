"""Torch restatement of the reference's temporal-aware DiT forward -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing
under gvfdiffusion_amd/ does.  It is a functional (state_dict in, tensor out) restatement of

    model/dit.py:449-480   DiT._forward
    model/dit.py:227-278   ModulatedSparseTransformerCrossBlock._forward
    model/dit.py:43-56     AbsolutePositionEmbedder.forward
    model/dit.py:72-100    TimestepEmbedder
    model/dit.py:298-303   FinalLayer.forward
    model/attention/modules.py:8-15,112-146   MultiHeadRMSNorm, MultiHeadAttention.forward
    model/attention/full_attn.py:23-35        softmax(q k^T / sqrt(d)) v

written independently of the product module, and PINNED by tests/golden/dit_small_golden.npz and
dit_full_golden.npz -- outputs of the reference's own model/dit.py imported in the build container
(tests/golden/make_golden.py).  Being torch code it runs on CPU (fp32 baseline) or, in GPU tests, on the
device as the checker for full-size shapes.

precision="fp32": the reference's eager fp32 semantics (what the goldens were produced with).
precision="bf16": same graph with the HIP pipeline's storage roundings inserted (bf16 operands of every
  big contraction with fp32 accumulation; bf16 q/k/v/P/attention-out/MLP-hidden; fp32 residual stream,
  LayerNorm, softmax, modulation, and the small projections listed in FP32_SITES: input / final layer, timestep
  embedder, adaLN projections, condition projections and to_kv) -- the "same-dtype oracle" BASELINE.json's DiT
  tolerance refers to.
"""
import math

import torch
import torch.nn.functional as F


_DT = {"bf16": torch.bfloat16, "fp16": torch.float16}


def _r(x, precision, site="gemm"):
    """Storage rounding of a tensor at a rounding point of the HIP pipeline.  precision: "fp32" (none), "bf16", "fp16", or a
    dict {"gemm": .., "attn": ..} giving the operand type of the projections' matrix products ("gemm": activations, weights, MLP
    hidden units, attention outputs) and of the attention's own operands ("attn": q, k, v, P) separately."""
    p = precision.get(site, precision.get("attn", "fp32") if site == "attn_pv" else "fp32") if isinstance(precision, dict) else precision
    return x if p == "fp32" else x.to(_DT[p]).to(torch.float32)


def _reduced(precision):
    return precision != "fp32"


# The HIP pipeline keeps the small projections in fp32 (csrc/elem.hip, and plain fp32 library GEMMs for the hoisted ones): 0.3 % of the FLOPs,
# more than a third of the bf16 error.  precision="bf16" follows that placement.
FP32_SITES = ("input_layer", "t_embedder.", "image_cond_proj", "static_cond_proj", "final_layer.", ".adaLN_modulation", ".to_kv")


def linear(x, sd, prefix, precision, round_out=False):
    w, b = sd[prefix + ".weight"], sd.get(prefix + ".bias")
    if _reduced(precision) and any(s in prefix or prefix.startswith(s) for s in FP32_SITES):
        y = F.linear(x, w, None)
        return y if b is None else y + b
    y = F.linear(_r(x, precision), _r(w, precision), None)
    if b is not None:
        y = y + b
    return _r(y, precision, "attn") if round_out else y          # the projections whose outputs are stored: q / qkv


def layer_norm(x, eps=1e-6):
    return F.layer_norm(x, x.shape[-1:], None, None, eps)


def rms_norm_heads(x, gamma, precision):
    """MultiHeadRMSNorm: normalize(x.float(), dim=-1) * gamma[H,d] * sqrt(d), cast back to x's dtype."""
    y = F.normalize(x.float(), dim=-1) * gamma * (x.shape[-1] ** 0.5)
    return _r(y, precision, "attn")


def sdpa(q, k, v, precision):
    """q [N,Lq,H,d], k/v [N,Lk,H,d] -> [N,Lq,H,d]."""
    q, k, v = (t.permute(0, 2, 1, 3) for t in (q, k, v))
    s = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(q.shape[-1]))
    p = torch.softmax(s, dim=-1)
    if _reduced(precision):
        # the kernel rounds the un-normalised probabilities to the operand type and divides by the fp32 row sum afterwards
        m = s.amax(dim=-1, keepdim=True)
        e = torch.exp(s - m)
        o = (_r(e, precision, "attn_pv") @ _r(v, precision, "attn_pv")) / e.sum(dim=-1, keepdim=True)
    else:
        o = p @ v
    return _r(o.permute(0, 2, 1, 3), precision, "gemm")


LOG2E = 1.4426950408889634


def sdpa_tiled(q, k, v, precision, gamma_k=None):
    """Cross attention with the rounding points of the tiled-cache kernel (gvfdiffusion_amd/csrc/attn_xt.hip):
    q [N,Lq,H,d] already rounded; k, v [N,Lk,H,d] are the UNROUNDED fp32 projections.  The cache builder folds the
    (optional) MultiHeadRMSNorm of k and the factor softmax_scale * log2(e) into k in fp32 and rounds ONCE; the kernel
    then takes P = bf16(exp2(q . k')) with no running maximum (softmax is shift invariant) and divides by the sum of the
    same bf16-rounded probabilities.  precision="fp32" is the plain softmax(q k^T / sqrt(d)) v."""
    d = q.shape[-1]
    if gamma_k is not None:
        k = F.normalize(k.float(), dim=-1) * gamma_k * (d ** 0.5)
    if not _reduced(precision):
        return sdpa(q, k, v, precision)
    q, k, v = (t.permute(0, 2, 1, 3) for t in (q, k, v))
    k2 = _r(k * (LOG2E / math.sqrt(d)), precision, "attn")
    s = q @ k2.transpose(-2, -1)
    if (precision.get("attn_pv", precision.get("attn")) if isinstance(precision, dict) else precision) == "fp16":
        # fp16 probabilities need a per-query shift (exp2 of a raw score overflows at 16): the kernel subtracts a shift through the
        # score accumulator's initial value; softmax is shift invariant, only the rounding grid of P moves with it
        s = s - s.amax(dim=-1, keepdim=True)
    p = _r(torch.exp2(s), precision, "attn_pv")
    # the kernel's range guard: a query whose denominator leaves (2^-100, 2^100) (or is not finite) sends its 256-query workgroup to the exact
    # running-maximum pass; emulated with the row maximum as the shift (the rounding grid of P moves with the shift, nothing else)
    l = p.sum(dim=-1)
    bad = ~((l > 2.0 ** -100) & (l < 2.0 ** 100))
    if bool(bad.any()):
        Lq = bad.shape[-1]
        blocks = (Lq + 255) // 256
        bad_blk = torch.nn.functional.pad(bad, (0, blocks * 256 - Lq)).reshape(*bad.shape[:-1], blocks, 256).any(dim=-1, keepdim=True)
        bad = bad_blk.expand(*bad.shape[:-1], blocks, 256).reshape(*bad.shape[:-1], blocks * 256)[..., :Lq]
        p_exact = _r(torch.exp2(s - s.amax(dim=-1, keepdim=True)), precision, "attn_pv")
        p = torch.where(bad[..., None], p_exact, p)
    o = (p @ _r(v, precision, "attn_pv")) / p.sum(dim=-1, keepdim=True)
    return _r(o.permute(0, 2, 1, 3), precision, "gemm")


def self_attention(x, sd, prefix, heads, precision, tiled=False):
    """tiled=True: the rounding points of the path the DiT's spatial self attention takes (K / V^T of the packed qkv
    projection re-tiled per step, csrc/attn_xt.hip); False: the streaming / one-wave kernels (csrc/attn.hip)."""
    B, L, C = x.shape
    qkv = linear(x, sd, prefix + ".to_qkv", precision, round_out=True).reshape(B, L, 3, heads, C // heads)
    q, k, v = qkv.unbind(dim=2)
    has_rms = prefix + ".q_rms_norm.gamma" in sd
    if has_rms:
        q = rms_norm_heads(q, sd[prefix + ".q_rms_norm.gamma"], precision)
    if tiled:
        h = sdpa_tiled(q, k, v, precision, gamma_k=sd[prefix + ".k_rms_norm.gamma"] if has_rms else None).reshape(B, L, C)
    else:
        if has_rms:
            k = rms_norm_heads(k, sd[prefix + ".k_rms_norm.gamma"], precision)
        h = sdpa(q, k, v, precision).reshape(B, L, C)
    return linear(h, sd, prefix + ".to_out", precision)


def cross_attention(x, ctx, sd, prefix, heads, precision):
    B, L, C = x.shape
    Lk = ctx.shape[1]
    q = linear(x, sd, prefix + ".to_q", precision, round_out=True).reshape(B, L, heads, C // heads)
    kv = linear(ctx, sd, prefix + ".to_kv", precision).reshape(B, Lk, 2, heads, C // heads)     # fp32: rounded by the cache builder
    k, v = kv.unbind(dim=2)
    gk = None
    if prefix + ".q_rms_norm.gamma" in sd:
        q = rms_norm_heads(q, sd[prefix + ".q_rms_norm.gamma"], precision)
        gk = sd[prefix + ".k_rms_norm.gamma"]
    h = sdpa_tiled(q, k, v, precision, gamma_k=gk).reshape(B, L, C)
    return linear(h, sd, prefix + ".to_out", precision)


def timestep_embedding(t, dim=256, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def absolute_position_embedding(xyz, channels):
    """(B,L,3) -> (B,L,channels): per axis [sin(x f), cos(x f)], f = 10000^(-i/F), F = channels//3//2; zero pad."""
    B, L, D = xyz.shape
    fd = channels // D // 2
    freqs = 1.0 / (10000 ** (torch.arange(fd, dtype=torch.float32, device=xyz.device) / fd))
    out = torch.outer(xyz.reshape(-1), freqs)
    emb = torch.cat([torch.sin(out), torch.cos(out)], dim=-1).reshape(B * L, -1)
    if emb.shape[1] < channels:
        emb = torch.cat([emb, torch.zeros(B * L, channels - emb.shape[1], device=xyz.device)], dim=-1)
    return emb.reshape(B, L, channels)


def modulate(h, shift, scale):
    return h * (1 + scale[:, None, None]) + shift[:, None, None]


def block_forward(x, mod, image_emb, static_emb, sd, p, heads, precision, no_temporal_attn=False):
    """One ModulatedSparseTransformerCrossBlock; x (B,T,N,C) fp32, mod (B,C), contexts (B,T,L,C).  no_temporal_attn: the block has no
    temporal sub-layer (model/dit.py:241-242, 253-260)."""
    B, T, N, C = x.shape
    silu = F.silu(mod)
    m6 = linear(silu, sd, p + ".adaLN_modulation.1", precision)
    sh_s, sc_s, g_s, sh_m, sc_m, g_m = m6.chunk(6, dim=1)
    if not no_temporal_attn:
        sh_t, sc_t, g_t = linear(silu, sd, p + ".adaLN_modulation_temporal.1", precision).chunk(3, dim=1)
    # spatial self attention over the N tokens of each frame
    h = modulate(layer_norm(x), sh_s, sc_s)
    h = self_attention(h.reshape(B * T, N, C), sd, p + ".spatial_self_attn", heads, precision, tiled=True).reshape(B, T, N, C)
    x = x + h * g_s[:, None, None]
    # temporal self attention over the T frames of each token
    if not no_temporal_attn:
        h = modulate(layer_norm(x), sh_t, sc_t).transpose(1, 2).reshape(B * N, T, C)
        h = self_attention(h, sd, p + ".temporal_self_attn", heads, precision).reshape(B, N, T, C).transpose(1, 2)
        x = x + h * g_t[:, None, None]
    # image cross attention (affine LayerNorm, no gate)
    h = F.layer_norm(x, (C,), sd[p + ".norm3.weight"], sd[p + ".norm3.bias"], 1e-6)
    h = cross_attention(h.reshape(B * T, N, C), image_emb.reshape(B * T, -1, C), sd, p + ".image_cross_attn", heads, precision)
    x = x + h.reshape(B, T, N, C)
    # static cross attention
    h = F.layer_norm(x, (C,), sd[p + ".norm4.weight"], sd[p + ".norm4.bias"], 1e-6)
    h = cross_attention(h.reshape(B * T, N, C), static_emb.reshape(B * T, -1, C), sd, p + ".static_cross_attn", heads, precision)
    x = x + h.reshape(B, T, N, C)
    # MLP
    h = modulate(layer_norm(x), sh_m, sc_m)
    h = linear(h, sd, p + ".mlp.mlp.0", precision)
    h = _r(F.gelu(h, approximate="tanh"), precision, "gemm")
    h = linear(h, sd, p + ".mlp.mlp.2", precision)
    return x + h * g_m[:, None, None]


def dit_forward(sd, cfg, x, t, cond_images, static_latent, deformation_position_xyz, precision="fp32",
                return_intermediates=False):
    """sd: state_dict (reference key names); cfg: configs/diffusion.yml `model:` dict."""
    assert isinstance(precision, dict) or precision in ("fp32", "bf16", "fp16")
    C, heads, nblocks = cfg["model_channels"], cfg["num_heads"], cfg["num_blocks"]
    B, T, N, _ = x.shape
    h = linear(x, sd, "input_layer", precision)
    t_emb = linear(F.silu(linear(timestep_embedding(t), sd, "t_embedder.mlp.0", precision)), sd, "t_embedder.mlp.2", precision)
    image_emb = linear(cond_images, sd, "image_cond_proj", precision, round_out=True)
    static_emb = linear(static_latent, sd, "static_cond_proj", precision, round_out=True)[:, None].expand(B, T, -1, C)
    assert cfg.get("pe_mode", "learnable") == "ape"
    h = h + absolute_position_embedding(deformation_position_xyz, C)[:, None]
    inter = {"h0": h, "t_emb": t_emb, "blocks": []}
    for i in range(nblocks):
        h = block_forward(h, t_emb, image_emb, static_emb, sd, f"blocks.{i}", heads, precision, no_temporal_attn=cfg.get("no_temporal_attn", False))
        if i == 0:
            inter["block0"] = h
        if return_intermediates:
            inter["blocks"].append(h)
    shift, scale = linear(F.silu(t_emb), sd, "final_layer.adaLN_modulation.1", precision).chunk(2, dim=1)
    h = modulate(layer_norm(h), shift, scale)
    y = linear(h, sd, "final_layer.linear", precision)
    return (y, inter) if return_intermediates else y
