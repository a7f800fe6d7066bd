"""Torch restatement of the motion-VAE decode -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows model/autoencoder.py: decode :579-609, process_chunk/chunk_forward :552-577, PreNorm :73-88,
Attention :109-163 (softmax(q k^T * dim_head^-0.5) v, to_out with bias, to_q/to_kv without), FeedForward/GEGLU
:90-107 (x * gelu(gates), exact erf GELU), PointEmbed :250-301, embeddings :392-394.  Pinned by
tests/golden/vae_small_golden.npz (outputs of the reference class imported in the build container)."""
import math

import torch
import torch.nn.functional as F


def _r(x, precision):
    return x.to(torch.bfloat16).to(torch.float32) if precision == "bf16" else x


def _lin(x, sd, name, precision, round_out=False):
    y = F.linear(_r(x, precision), _r(sd[name + ".weight"], precision))
    if name + ".bias" in sd:
        y = y + sd[name + ".bias"]
    return _r(y, precision) if round_out else y


def _ln(x, eps=1e-6):   # PreNorm :77 eps 1e-6; the embedding norms :393-394 keep nn.LayerNorm's default 1e-5
    return F.layer_norm(x, x.shape[-1:], None, None, eps)


def attention(xq, ctx, sd, prefix, heads, precision):
    B, N, C = xq.shape
    q = _lin(xq, sd, prefix + ".to_q", precision, True)
    k, v = _lin(ctx, sd, prefix + ".to_kv", precision, True).chunk(2, dim=-1)
    d = q.shape[-1] // heads
    q, k, v = (t.reshape(t.shape[0], t.shape[1], heads, d).permute(0, 2, 1, 3) for t in (q, k, v))
    s = (q @ k.transpose(-1, -2)) * d ** -0.5
    if precision == "bf16":
        e = torch.exp(s - s.amax(dim=-1, keepdim=True))
        o = (_r(e, precision) @ v) / e.sum(dim=-1, keepdim=True)
    else:
        o = torch.softmax(s, dim=-1) @ v
    o = _r(o.permute(0, 2, 1, 3).reshape(B, N, -1), precision)
    return _lin(o, sd, prefix + ".to_out", precision)


def feed_forward(x, sd, prefix, precision):
    h = _lin(x, sd, prefix + ".net.0", precision, True)
    a, gates = h.chunk(2, dim=-1)
    return _lin(_r(a * F.gelu(gates), precision), sd, prefix + ".net.2", precision)


def point_embed(xyz, omega):
    emb = [torch.cat([torch.sin(xyz[..., k:k + 1].double() * omega), torch.cos(xyz[..., k:k + 1].double() * omega)], dim=-1)
           for k in range(3)]
    return torch.cat(emb, dim=-1).to(xyz.dtype)


def vae_decode(sd, cfg, x, queries, num_timesteps, precision="fp32"):
    """x (B*T, L, latent_dim), queries (B, P, 14) -> (B, T, P, output_dim)."""
    heads, depth = cfg["heads"], cfg["depth"]
    B, P = queries.shape[:2]
    T = num_timesteps
    h = _lin(x, sd, "proj", precision)
    for i in range(depth):
        h = attention(_ln(h), _ln(h), sd, f"layers.{i}.0.fn", heads, precision) + h
        h = feed_forward(_ln(h), sd, f"layers.{i}.1.fn", precision) + h
    q_embed = _ln(_lin(queries, sd, "gs_embedding.0", precision), 1e-5) + _ln(point_embed(queries[..., :3], sd["position_encoding.0.omega"]), 1e-5)
    q_embed = q_embed[:, None].expand(B, T, P, -1).reshape(B * T, P, -1)
    lat = attention(_ln(q_embed), _ln(h), sd, "decoder_cross_attn.fn", heads, precision)
    return _lin(lat, sd, "to_outputs", precision).reshape(B, T, P, -1)
