"""Torch restatement of the motion-VAE decode -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows model/autoencoder.py: decode :579-609, process_chunk/chunk_forward :552-577, PreNorm :73-88,
Attention :109-163 (softmax(q k^T * dim_head^-0.5) v, to_out with bias, to_q/to_kv without), FeedForward/GEGLU
:90-107 (x * gelu(gates), exact erf GELU), PointEmbed :250-301, embeddings :392-394.  Pinned by
tests/golden/vae_small_golden.npz (outputs of the reference class imported in the build container)."""
import torch
import torch.nn.functional as F


_DT = {"bf16": torch.bfloat16, "fp16": torch.float16}


def _r(x, precision):
    """storage rounding at a rounding point of the HIP pipeline: "fp32" (none), "bf16" or "fp16" (the reference's autocast dtype)"""
    return x if precision == "fp32" else x.to(_DT[precision]).to(torch.float32)


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
    if precision != "fp32":
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


# ---- encode (model/autoencoder.py:449-550) ------------------------------------------------------------------------
def delta_interp(static_gs, micro_static_pc, micro_moving_pc, knn_k=8, beta=7.0):
    """compute_delta_interp :449-500 with pytorch3d.ops.knn_points restated as a brute-force K-nearest search
    (squared distances ascending).  (B,L,3), (B,N,3), (B,T,N,3) -> (B,T,L,3)."""
    d2 = ((static_gs[:, :, None, :] - micro_static_pc[:, None, :, :]) ** 2).sum(-1)
    knn_dists, knn_idx = torch.topk(d2, knn_k, dim=-1, largest=False, sorted=True)
    radii = knn_dists.mean(dim=-1).sqrt() + 1e-6
    w = torch.exp(-beta * knn_dists / radii[..., None] ** 2) * (knn_dists <= radii[..., None] ** 2).float()
    w = w / (w.sum(dim=-1, keepdim=True) + 1e-8)
    B, L, K = knn_idx.shape
    T = micro_moving_pc.shape[1]
    out = torch.zeros((B, T, L, 3), dtype=static_gs.dtype)
    for b in range(B):
        nb0 = micro_static_pc[b][knn_idx[b]]                      # (L, K, 3)
        for t in range(T):
            out[b, t] = ((micro_moving_pc[b, t][knn_idx[b]] - nb0) * w[b][..., None]).sum(dim=1)
    return out


def vae_encode(sd, cfg, static_pc, delta_pc, input_static_gs, knn_k=8, beta=7.0, precision="fp32"):
    """-> (mean, logvar, estimated deltas); input_static_gs (B, L, 3) = xyz of the FPS-sampled Gaussians.  The point
    embeddings are fp32 in every precision (as in the HIP kernel); GEMM / attention operands are rounded for "bf16"."""
    heads = cfg["heads"]
    B, T = delta_pc.shape[:2]
    L, N = input_static_gs.shape[1], static_pc.shape[1]
    moving = delta_pc + static_pc[:, None]
    est = delta_interp(input_static_gs, static_pc, moving, knn_k, beta)
    om = sd["position_encoding.0.omega"]

    def embed(delta, xyz):
        return _ln(_lin(delta, sd, "input_embedding.0", "fp32"), 1e-5) + _ln(point_embed(xyz, om), 1e-5)[:, None]

    xq = embed(est, input_static_gs).reshape(B * T, L, -1)
    ctx = embed(delta_pc, static_pc).reshape(B * T, N, -1)
    x = attention(_ln(xq), _ln(ctx), sd, "cross_attend_blocks.0.fn", heads, precision) + xq
    x = feed_forward(_ln(x), sd, "cross_attend_blocks.1.fn", precision) + x
    return _lin(x, sd, "mean_fc", precision), _lin(x, sd, "logvar_fc", precision), est
