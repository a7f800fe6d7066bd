"""Torch restatement of the static-VAE backbone (SparseTransformerVAE) -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows model/sparse_voxel_diffusion/sparse_transformer_vae.py: encode :158-182, decode :184-196; the block
sparse_transformer.py:165-191 (unmodulated branch: x + attn(LN(x)); x + mlp(LN(x)), LayerNorm eps 1e-6, no affine);
the absolute position embedder :73-112; the swin window partition sparse/attention/windowed_attn.py:20-60 (coords + shift,
// window, tokens grouped by (batch, wx, wy, wz)); the qkv channel layouts of sparse/attention/modules.py:150-162; softmax
attention inside a window with scale head_dim^-0.5 (flash_attn_varlen_qkvpacked_func, a third-party kernel -- restated
from its definition); tanh-GELU MLP :115-126.  Operates on (feats (T, C), coords (T, 4) = batch, x, y, z).
Pinned by tests/golden/sparse_vae_golden.npz (outputs of the reference classes imported in the build container).

precision "bf16" / "fp16" rounds where the HIP path rounds: every GEMM / attention operand (LayerNorm outputs, q k v, the
probabilities' numerators, the attention output, the GELU output) to that type, fp32 everywhere else."""
import torch
import torch.nn.functional as F


_LP = {"bf16": torch.bfloat16, "fp16": torch.float16}


def _r(x, precision):
    return x.to(_LP[precision]).to(torch.float32) if precision in _LP else x


def _lin(x, sd, name, precision):
    return F.linear(_r(x, precision), _r(sd[name + ".weight"], precision)) + sd[name + ".bias"]


def ape(xyz, channels):
    freq_dim = channels // 3 // 2
    freqs = 1.0 / (10000 ** (torch.arange(freq_dim, dtype=torch.float32) / freq_dim))
    out = torch.outer(xyz.reshape(-1).float(), freqs)
    e = torch.cat([torch.sin(out), torch.cos(out)], dim=-1).reshape(xyz.shape[0], -1)
    if e.shape[1] < channels:
        e = torch.cat([e, torch.zeros(xyz.shape[0], channels - e.shape[1])], dim=-1)
    return e


def window_ids(coords, window, shift):
    """One integer per token identifying its (batch, window): windowed_attn.py:34-47."""
    c = coords.long().clone()
    c[:, 1:] += shift
    n = [int(c[:, 1 + a].max()) // window + 1 for a in range(3)]
    w = c[:, 1:] // window
    return ((c[:, 0] * n[0] + w[:, 0]) * n[1] + w[:, 1]) * n[2] + w[:, 2]


def attention_groups(q, k, v, gid, precision):
    """q k v (T, H, d); softmax attention among the tokens sharing a group id."""
    out = torch.zeros_like(q)
    d = q.shape[-1]
    for g in torch.unique(gid):
        idx = torch.nonzero(gid == g).squeeze(1)
        qg, kg, vg = (t[idx].permute(1, 0, 2) for t in (q, k, v))                 # (H, n, d)
        s = (qg @ kg.transpose(-1, -2)) * d ** -0.5
        if precision in _LP:
            e = torch.exp(s - s.amax(dim=-1, keepdim=True))
            o = (_r(e, precision) @ vg) / e.sum(dim=-1, keepdim=True)
        else:
            o = torch.softmax(s, dim=-1) @ vg
        out[idx] = o.permute(1, 0, 2)
    return out


def block(x, gid, sd, prefix, heads, precision, old_impl, qk_rms_norm=False):
    T, C = x.shape
    d = C // heads
    h = _r(F.layer_norm(x, (C,), eps=1e-6), precision)
    qkv = _r(_lin(h, sd, prefix + ".attn.to_qkv", precision), precision)
    if old_impl:
        q, k, v = qkv.reshape(T, heads, 3 * d).chunk(3, dim=-1)
    else:
        q, k, v = qkv.reshape(T, 3, heads, d).unbind(dim=1)
    if qk_rms_norm:   # SparseMultiHeadRMSNorm, trellis/modules/sparse/attention/modules.py:11-25: normalize * gamma * sqrt(d)
        q = _r(F.normalize(q, dim=-1) * sd[prefix + ".attn.q_rms_norm.gamma"] * d ** 0.5, precision)
        k = _r(F.normalize(k, dim=-1) * sd[prefix + ".attn.k_rms_norm.gamma"] * d ** 0.5, precision)
    a = _r(attention_groups(q, k, v, gid, precision).reshape(T, C), precision)
    x = x + _lin(a, sd, prefix + ".attn.to_out", precision)
    h = _r(F.layer_norm(x, (C,), eps=1e-6), precision)
    h = _r(F.gelu(_lin(h, sd, prefix + ".mlp.mlp.0", precision), approximate="tanh"), precision)
    return x + _lin(h, sd, prefix + ".mlp.mlp.2", precision)


def _torso(rows, coords, sd, cfg, first, stack, last, precision):
    C, heads, window = cfg["model_channels"], cfg["num_heads"], cfg["window_size"]
    assert cfg.get("attn_mode", "swin") == "swin" and cfg.get("pe_mode", "ape") == "ape"
    x = _lin(rows, sd, first, precision) + ape(coords[:, 1:], C)
    for i in range(cfg["num_blocks"]):
        gid = window_ids(coords, window, window // 2 * (i % 2))
        x = block(x, gid, sd, f"{stack}.{i}", heads, precision, cfg.get("use_old_attn_impl", True))
    if cfg.get("norm_output", False):
        x = F.layer_norm(x, (C,))
    return _lin(x, sd, last, precision)


def encode(sd, cfg, feats, coords, precision="fp32"):
    """-> (mean, logvar), each (T, latent_channels)."""
    return _torso(feats, coords, sd, cfg, "input_layer", "encoder", "to_latent", precision).chunk(2, dim=-1)


def decode(sd, cfg, z, coords, precision="fp32"):
    return _torso(z, coords, sd, cfg, "from_latent", "decoder", "out_layer", precision)


def slat_decode_rows(sd, cfg, feats, coords, precision="fp32"):
    """TRELLIS SLatGaussianDecoder up to its output rows (trellis/models/structured_latent_vae/base.py:108-117,
    decoder_gs.py:117-121): input layer + APE, swin blocks ([q|k|v][head][c] layout, optional QK-RMSNorm), LayerNorm, out_layer."""
    C, heads, window = cfg["model_channels"], cfg["num_heads"], cfg["window_size"]
    x = _lin(feats, sd, "input_layer", precision) + ape(coords[:, 1:], C)
    for i in range(cfg["num_blocks"]):
        gid = window_ids(coords, window, window // 2 * (i % 2))
        x = block(x, gid, sd, f"blocks.{i}", heads, precision, False, cfg.get("qk_rms_norm", False))
    return _lin(F.layer_norm(x, (C,)), sd, "out_layer", precision)
