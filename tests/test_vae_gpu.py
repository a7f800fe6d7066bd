"""Motion-VAE decode on the HIP kernels (gvfdiffusion_amd/model/autoencoder.py) against the torch oracle
(oracle/vae_ref.py, pinned bit-exactly to the reference by tests/test_oracle_vae.py).

Tolerances (relative L2 over the whole output, as for the DiT):
  vs the same-dtype oracle   6e-3 (bf16) / 9e-4 (fp16)   (same rounding points; differences = accumulation order, exp2 softmax,
                                                          folded to_out∘to_outputs and the fp32 query embedding)
  vs the fp32 oracle         8.5e-3 (bf16) / 1e-3 (fp16) (operand rounding through `depth` blocks); measured values + 30 %
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

TOL_VS_BF16_ORACLE = 6e-3        # measured 4.4e-3 .. 4.5e-3 (decode, three configs), + 30 %
TOL_VS_FP32_ORACLE = 8.5e-3      # measured 5.6e-3 .. 6.2e-3
# fp16 operands (set_compute_dtype / an fp16 autocast region: what the reference decodes under, inference_dpm_latent.py:256-257)
TOL_FP16_VS_FP16_ORACLE = 9e-4   # measured 4.7e-4 .. 6.5e-4
TOL_FP16_VS_FP32_ORACLE = 1e-3   # measured 6.5e-4 .. 7.2e-4: 9 x closer to the fp32 reference than bf16 operands


def _model(cfg, seed=0, gain=1.0):
    from gvfdiffusion_amd.model.autoencoder import GSKLTemporalVariationalAutoEncoder
    torch.manual_seed(seed)
    m = GSKLTemporalVariationalAutoEncoder(**cfg)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if p.dim() == 2:
                p.copy_(torch.randn_like(p) * (gain / math.sqrt(p.shape[1])))
            else:
                p.copy_(torch.randn_like(p) * 0.1)
    return m


def _inputs(cfg, B, P, L, seed=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B * cfg["num_timesteps"], L, cfg["latent_dim"], generator=g)
    q = torch.randn(B, P, 14, generator=g)
    q[..., :3] = torch.rand(B, P, 3, generator=g) - 0.5
    return x, q


def _rel(a, b):
    return float((a - b).norm() / b.norm())


def _check(cfg, B, P, L, chunk_rows=None, gain=1.0, dtype="bf16"):
    from oracle import vae_ref
    m = _model(cfg, gain=gain)
    x, q = _inputs(cfg, B, P, L)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    from oracle_cache import oracle_cached
    with torch.no_grad():          # (the oracle's results are cached on disk by content: the fp32 one serves both operand types of a configuration)
        ref32 = oracle_cached("vae_decode", vae_ref, (cfg, sd, x, q, "fp32"), lambda: vae_ref.vae_decode(sd, cfg, x, q, cfg["num_timesteps"], "fp32"))
        ref16 = oracle_cached("vae_decode", vae_ref, (cfg, sd, x, q, dtype), lambda: vae_ref.vae_decode(sd, cfg, x, q, cfg["num_timesteps"], dtype))
    m = m.cuda().set_compute_dtype(dtype)
    if chunk_rows:
        m.max_chunk_rows = chunk_rows
    y = m.decode(x.cuda(), q.cuda()).cpu()
    assert m._wcache["lp"] == {"bf16": torch.bfloat16, "fp16": torch.float16}[dtype]
    assert y.shape == (B, cfg["num_timesteps"], P, cfg["output_dim"]) and torch.isfinite(y).all()
    e16, e32 = _rel(y, ref16), _rel(y, ref32)
    print(f"vae decode [{dtype}] rel-L2: vs {dtype} oracle {e16:.2e}, vs fp32 oracle {e32:.2e}, {dtype} oracle vs fp32 oracle {_rel(ref16, ref32):.2e}")
    t16, t32 = (TOL_VS_BF16_ORACLE, TOL_VS_FP32_ORACLE) if dtype == "bf16" else (TOL_FP16_VS_FP16_ORACLE, TOL_FP16_VS_FP32_ORACLE)
    assert e16 < t16 and e32 < t32, (e16, e32)
    return e16, e32


BASE = dict(depth=2, dim=192, queries_dim=192, output_dim=14, num_inputs=64, num_latents=40, latent_dim=16, heads=3,
            dim_head=-1, num_timesteps=3, chunk_size=100)


def test_geglu_matches_torch(cuda):
    from gvfdiffusion_amd.ops import vae_ops
    torch.manual_seed(0)
    x = (torch.randn(777, 2 * 264) * 2).to(torch.bfloat16).cuda()
    y = vae_ops.geglu_bf16(x).float().cpu()
    a, g = x.float().cpu().chunk(2, dim=-1)
    ref = (a * torch.nn.functional.gelu(g)).to(torch.bfloat16).float()
    assert (y - ref).abs().max() <= 2 ** -7 * ref.abs().max()      # at most one bf16 ulp of the largest value
    assert ((y - ref).abs() > 0).float().mean() < 1e-2             # ... and on <1% of the elements (erff vs torch erf)


@pytest.mark.parametrize("C", [96, 192, 768])
def test_query_embed_matches_torch(cuda, C):
    from gvfdiffusion_amd.ops import vae_ops
    from oracle import vae_ref
    torch.manual_seed(C)
    P = 1000
    q = torch.randn(P, 14)
    q[:, :3] = torch.rand(P, 3) - 0.5
    w, b = torch.randn(C, 14) * 0.3, torch.randn(C) * 0.1
    E = C // 6
    omega = 1.0 / 10000 ** (torch.arange(E, dtype=torch.float64) / (E / 2.0))
    ref = vae_ref._ln(vae_ref._ln(q @ w.T + b, 1e-5) + vae_ref._ln(vae_ref.point_embed(q[:, :3], omega), 1e-5))
    y = vae_ops.vae_query_embed_bf16(q.cuda(), w.cuda(), b.cuda(), omega.float().cuda()).float().cpu()
    assert (y - ref).abs().max() < 2e-2 and _rel(y, ref) < 3e-3     # bf16 output rounding: 2^-9 relative


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_decode_head_dim_64_chunked(cuda, dtype):
    _check(BASE, B=2, P=300, L=40, chunk_rows=6 * 128, dtype=dtype)     # 3 chunks of 128 Gaussians, the last ragged (44)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_decode_head_dim_32(cuda, dtype):
    _check(dict(BASE, heads=6), B=1, P=257, L=64, dtype=dtype)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_decode_released_config(cuda, dtype):
    """depth 12, dim 768, 12 heads of 64, 512 latents, 24 frames (configs/diffusion.yml: autoencoder section)."""
    cfg = dict(depth=12, dim=768, queries_dim=768, output_dim=14, num_inputs=8192, num_latents=512, latent_dim=16, heads=12,
               dim_head=-1, num_timesteps=24, chunk_size=8192)
    _check(cfg, B=1, P=1500, L=512, dtype=dtype)


def test_decode_follows_an_autocast_region(cuda):
    """`with accelerator.autocast(): vae.decode(...)` (inference_dpm_latent.py:256-257, mixed_precision='fp16') selects fp16 operands."""
    m = _model(BASE).cuda()
    x, q = _inputs(BASE, 1, 130, 40)
    y_default = m.decode(x.cuda(), q.cuda())
    assert m._wcache["lp"] == torch.bfloat16
    with torch.autocast("cuda", dtype=torch.float16):
        y16 = m.decode(x.cuda(), q.cuda())
    assert m._wcache["lp"] == torch.float16 and y16.dtype == torch.float32 and not torch.equal(y16, y_default)


def test_strict_load_of_reference_layout_and_loud_failures(cuda):
    import json, os
    from gvfdiffusion_amd._lib import GvfError
    from gvfdiffusion_amd.model.autoencoder import GSKLTemporalVariationalAutoEncoder
    man = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vae_manifest.json")))
    cfg = dict(man["config"])
    m = GSKLTemporalVariationalAutoEncoder(**cfg)
    sd = {k: torch.zeros(s, dtype=torch.float64 if k.endswith("omega") else torch.float32) for k, s in man["state_dict"].items()}
    m.load_state_dict(sd, strict=True)
    with pytest.raises(GvfError):
        m.decode(torch.zeros(24, 512, 16), torch.zeros(1, 8, 14))          # CPU tensors: no fallback
    with pytest.raises(GvfError):
        m.encode(torch.zeros(1, 8192, 3), torch.zeros(1, 24, 8192, 3), [torch.zeros(2000, 14)])   # CPU tensors: no fallback


# ---- encode (model/autoencoder.py:502-550) -------------------------------------------------------------------------
def _encode_inputs(cfg, B, n_gs, seed=2):
    g = torch.Generator().manual_seed(seed)
    T, N = cfg["num_timesteps"], cfg["num_inputs"]
    static_pc = torch.rand((B, N, 3), generator=g) - 0.5
    delta_pc = torch.randn((B, T, N, 3), generator=g) * 0.05
    gs = [torch.rand((n, 14), generator=g) - 0.5 for n in n_gs]
    return static_pc, delta_pc, gs


def _check_encode(cfg, B, n_gs):
    import numpy as np
    from oracle import points_ref, vae_ref
    m = _model(cfg, seed=3)
    static_pc, delta_pc, gs = _encode_inputs(cfg, B, n_gs)
    L = cfg["num_latents"]
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    ptr = np.concatenate([[0], np.cumsum(n_gs)]).tolist()
    idx = points_ref.fps_indices(np.concatenate([t[:, :3].numpy() for t in gs]), ptr, [L] * B, [0] * B)
    sampled_ref = torch.cat(gs)[torch.from_numpy(idx)].reshape(B, L, 14)
    with torch.no_grad():
        ref32 = vae_ref.vae_encode(sd, cfg, static_pc, delta_pc, sampled_ref[..., :3].contiguous(), m.knn_k, m.beta, "fp32")
        ref16 = vae_ref.vae_encode(sd, cfg, static_pc, delta_pc, sampled_ref[..., :3].contiguous(), m.knn_k, m.beta, "bf16")
    m = m.cuda()
    kl, z, post, sampled = m.encode(static_pc.cuda(), delta_pc.cuda(), [t.cuda() for t in gs], random_start=False, sample_posterior=False)
    assert torch.equal(sampled.cpu(), sampled_ref)                         # farthest point sampling: index-exact
    est = m.compute_delta_interp(sampled[..., :3].contiguous(), static_pc.cuda(), (delta_pc + static_pc[:, None]).cuda(), m.knn_k, m.beta)
    assert (est.cpu() - ref32[2]).abs().max() < 1e-6
    T = cfg["num_timesteps"]
    assert post.mean.shape == (B * T, L, cfg["latent_dim"]) and torch.equal(z, post.mean) and kl.shape == (B * T,)
    for got, r16, r32, name in ((post.mean.cpu(), ref16[0], ref32[0], "mean"), (post.logvar.cpu(), ref16[1], ref32[1], "logvar")):
        e16, e32 = _rel(got, r16), _rel(got, r32)
        print(f"vae encode {name} rel-L2: vs bf16 oracle {e16:.2e}, vs fp32 oracle {e32:.2e}")
        assert e16 < TOL_VS_BF16_ORACLE and e32 < TOL_VS_FP32_ORACLE, (name, e16, e32)
    kl_ref = 0.5 * (ref32[0] ** 2 + ref32[1].exp() - 1.0 - ref32[1]).mean(dim=(1, 2))
    assert _rel(kl.cpu(), kl_ref) < TOL_VS_FP32_ORACLE
    return m, (static_pc, delta_pc, gs)


def test_encode_head_dim_64(cuda):
    _check_encode(BASE, B=2, n_gs=(150, 97))


def test_encode_head_dim_32_ragged_context(cuda):
    _check_encode(dict(BASE, heads=6, num_inputs=200, num_latents=64), B=1, n_gs=(333,))


def test_encode_samples_the_posterior_and_round_trips_through_decode(cuda):
    m, (static_pc, delta_pc, gs) = _check_encode(BASE, B=2, n_gs=(150, 97))
    torch.manual_seed(0)
    kl, z, post, sampled = m.encode(static_pc.cuda(), delta_pc.cuda(), [t.cuda() for t in gs])
    assert sampled.shape == (2, BASE["num_latents"], 14) and torch.isfinite(z).all()
    zs = (z - post.mean) / post.std                                        # standard normal draws
    assert abs(float(zs.mean())) < 0.1 and abs(float(zs.std()) - 1.0) < 0.1
    out = m([t.cuda() for t in gs], static_pc.cuda(), delta_pc.cuda())
    assert out["logits"].shape == (2, BASE["num_timesteps"], 150, BASE["output_dim"]) and torch.isfinite(out["logits"]).all()


def test_kv_resident_attention_variant_forced_everywhere(cuda):
    """csrc/attn.hip has a K/V-resident variant for key sets <= 512 that by default only the decoder's cross attention takes;
    GVF_ATTN_KVRES=2 forces it for every eligible call (head_dim 32 and 64, row-major and transposed V, RMS-normed q / k,
    varlen windows).  The switch is read once per process, so the parity tests are re-run in a child process with it set."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GVF_ATTN_KVRES="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_dit_gpu.py::test_attention_matches_oracle",
                        "tests/test_dit_gpu.py::test_attention_operator_call_forms_and_strided_views",
                        "tests/test_vae_gpu.py::test_decode_head_dim_64_chunked", "tests/test_vae_gpu.py::test_decode_head_dim_32",
                        "tests/test_vae_gpu.py::test_encode_head_dim_64", "tests/test_sparse.py", "tests/test_sparse_vae_gpu.py::test_released_width_ragged_batch"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("lp", [torch.bfloat16, torch.float16])
def test_kv_resident_attention_survives_scores_that_overflow_exp2(cuda, lp):
    """The K/V-resident kernel (decoder cross attention: head_dim 64, transposed V, 512 keys, >= 1024 queries) computes P = exp2(s)
    without a running maximum and checks every denominator: a query whose scores exceed the exponent range must come out of the exact
    running-maximum pass, its neighbours in other waves unaffected.  fp16: the max-free pass shifts every query by its best score against
    the first key tile; the spiked queries overflow fp16 (or underflow everything else) and take the exact pass the same way."""
    from gvfdiffusion_amd.ops import dit_ops
    g = torch.Generator().manual_seed(11)
    Lq, Lk, H, D = 2048, 512, 2, 64
    q = torch.randn((1, Lq, H, D), generator=g)
    k = torch.randn((1, Lk, H, D), generator=g)
    v = torch.randn((1, Lk, H, D), generator=g)
    q[0, 5] *= 400.0                                           # |s| ~ 400 * 8 / sqrt(64) * log2 e: exp2 overflows
    q[0, 1500, 1] *= -300.0
    qb, kb, vb = (t.to(lp).to(cuda) for t in (q, k, v))
    vt = vb.permute(0, 2, 3, 1).contiguous()                   # [1][H][D][Lk]: keys contiguous
    out = torch.empty_like(qb)
    sq, sk = (Lq * H * D, 0, H * D, D), (Lk * H * D, 0, H * D, D)
    dit_ops.attention_bf16(qb, kb, vt, out, 1, 1, Lq, Lk, H, sq, sk, (H * D * Lk, 0, Lk, D * Lk), sq, v_transposed=True, head_dim=D)
    s = torch.einsum("blhd,bmhd->bhlm", qb.float(), kb.float()) * D ** -0.5
    ref = torch.einsum("bhlm,bmhd->blhd", torch.softmax(s, dim=-1), vb.float())
    assert torch.isfinite(out).all()
    err = (out.float() - ref).norm(dim=-1) / ref.norm(dim=-1).clamp_min(1e-6)
    tol = 1.0 if lp == torch.bfloat16 else 0.15
    assert float(err[0, 5].max()) < 2e-2 * tol and float(err[0, 1500].max()) < 2e-2 * tol and float(err.max()) < 3e-2 * tol, (err[0, 5], err.max())


# ---- csrc/attn_xt64.hip: the decoder cross attention against the pre-tiled, LDS-resident latent set (head_dim 64) ----------------------

def _randomise(m):
    """every matrix ~ N(0, 1 / fan_in), every vector ~ N(0, 0.1): the constructor zero-initialises the output layer as the reference does"""
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * (p.shape[1] ** -0.5 if p.dim() == 2 else 0.1))


def _tiled64_case(cuda, lp, B, T, Lq, Lk, H, seed, spike=None, force_exact=False):
    from gvfdiffusion_amd.ops import dit_ops
    D, C = 64, H * 64
    g = torch.Generator().manual_seed(seed)
    q = torch.randn((B, Lq, C), generator=g)
    kv = torch.randn((B * T * Lk, 2 * C), generator=g)
    if spike is not None:
        for (b, row, h, gain) in spike:
            q[b, row, h * D:(h + 1) * D] *= gain
    qb, kvb = q.to(lp).to(cuda), kv.to(lp).to(cuda)
    kt, vt = dit_ops.attention_pack_kv64(kvb, B * T, Lk, H, 0, C)
    out = torch.full((B, T, Lq, C), float("nan"), dtype=lp, device=cuda)
    fb = torch.zeros(1, dtype=torch.int32, device=cuda)
    dit_ops.attention_tiled64(qb, kt, vt, out, B, T, Lq, Lk, H, (Lq * C, 0, C), (T * Lq * C, Lq * C, C), T, 1, force_exact=force_exact, fallback_counter=fb)
    k = kvb[:, :C].float().view(B, T, Lk, H, D)
    v = kvb[:, C:].float().view(B, T, Lk, H, D)
    s = torch.einsum("blhd,btmhd->bthlm", qb.float().view(B, Lq, H, D), k) * D ** -0.5
    ref = torch.einsum("bthlm,btmhd->btlhd", torch.softmax(s, dim=-1), v).reshape(B, T, Lq, C)
    return out, ref, int(fb.item())


@pytest.mark.parametrize("lp", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(1, 3, 2048 + 300, 512, 12), (2, 2, 777, 200, 3), (1, 1, 64, 64, 1), (1, 2, 5000, 40, 2), (1, 1, 1, 512, 2)])
def test_tiled64_attention_matches_fp32_softmax(cuda, lp, shape):
    """gvf_attn_pack_kv64 + gvf_attn_tiled64_fwd against softmax(q k^T / 8) v in fp32 on the same 16-bit operands: full and ragged query
    blocks (one workgroup covers 2048 queries in 8 passes of 256), key sets that end inside a tile, a single tile, one query; every
    output row written (the buffer starts as NaN), no wave on the exact path.  Bars = the probabilities' 16-bit rounding (bf16 2^-9
    relative per term, averaged over the keys; fp16 2^-12)."""
    B, T, Lq, Lk, H = shape
    out, ref, fb = _tiled64_case(cuda, lp, B, T, Lq, Lk, H, seed=5)
    assert torch.isfinite(out).all() and fb == 0
    rel = float((out.float() - ref).norm() / ref.norm())
    assert rel < (4e-3 if lp == torch.bfloat16 else 6e-4), rel
    assert float((out.float() - ref).abs().max()) < (3e-2 if lp == torch.bfloat16 else 4e-3)


@pytest.mark.parametrize("lp", [torch.bfloat16, torch.float16])
def test_tiled64_attention_exact_path_and_range_guard(cuda, lp):
    """force_exact runs every wave through the running-maximum softmax (same bars); queries whose scores leave the exponent range send THEIR
    wave pass (64 queries) to it and nobody else: the counter says how many."""
    out, ref, fb = _tiled64_case(cuda, lp, 1, 2, 2048 + 128, 512, 2, seed=6, force_exact=True)
    assert fb == 2 * 2 * ((2048 + 128) // 64)
    assert float((out.float() - ref).norm() / ref.norm()) < (4e-3 if lp == torch.bfloat16 else 6e-4)
    spikes = [(0, 5, 0, 400.0), (0, 1500, 1, -300.0)]
    out, ref, fb = _tiled64_case(cuda, lp, 1, 2, 2048 + 128, 512, 2, seed=7, spike=spikes)
    assert torch.isfinite(out).all()
    assert 1 <= fb <= 2 * 2, fb          # (query row, head) x 2 frames: one 64-query wave pass each; the shift may keep fp16's negative spike in range
    err = (out.float() - ref).view(1, 2, -1, 2, 64).norm(dim=-1) / ref.view(1, 2, -1, 2, 64).norm(dim=-1).clamp_min(1e-6)
    tol = 1.0 if lp == torch.bfloat16 else 0.15
    assert float(err[0, :, 5, 0].max()) < 2e-2 * tol and float(err[0, :, 1500, 1].max()) < 2e-2 * tol and float(err.max()) < 3e-2 * tol


def test_tiled64_entry_point_refuses_what_it_cannot_run(cuda):
    from gvfdiffusion_amd import _lib
    from gvfdiffusion_amd.ops import dit_ops
    q = torch.zeros((1, 64, 64), dtype=torch.bfloat16, device=cuda)
    kv = torch.zeros((576, 128), dtype=torch.bfloat16, device=cuda)
    kt, vt = dit_ops.attention_pack_kv64(kv, 1, 576, 1, 0, 64)
    out = torch.empty((1, 1, 64, 64), dtype=torch.bfloat16, device=cuda)
    with pytest.raises(_lib.GvfError):                 # 576 keys do not fit the LDS-resident set
        dit_ops.attention_tiled64(q, kt, vt, out, 1, 1, 64, 576, 1, (4096, 0, 64), (4096, 4096, 64), 1, 1)
    with pytest.raises(_lib.GvfError):                 # output rows must be 16-byte aligned
        dit_ops.attention_tiled64(q, kt, vt, out, 1, 1, 64, 512, 1, (4096, 0, 64), (4096, 4096, 68), 1, 1)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_decode_is_the_same_through_either_attention_kernel(cuda, dtype):
    """decode() through csrc/attn_xt64.hip (default) and through csrc/attn.hip's K/V-resident kernel (GVF_VAE_TILED64=0): same operands, same
    placement, two summation orders."""
    import os
    from gvfdiffusion_amd.model.autoencoder import GSKLTemporalVariationalAutoEncoder
    torch.manual_seed(3)
    m = GSKLTemporalVariationalAutoEncoder(depth=2, dim=384, queries_dim=384, output_dim=14, num_inputs=512, num_latents=128, latent_dim=16, heads=6,
                                           dim_head=64, num_timesteps=3)
    _randomise(m)
    m = m.to(cuda).set_compute_dtype(dtype)
    x = torch.randn(3, 128, 16, device=cuda)
    qs = torch.randn(1, 3000, 14, device=cuda)
    y1 = m.decode(x, qs)
    os.environ["GVF_VAE_TILED64"] = "0"
    try:
        y0 = m.decode(x, qs)
    finally:
        del os.environ["GVF_VAE_TILED64"]
    assert float((y1 - y0).norm() / y0.norm()) < (3e-3 if dtype == "bf16" else 4e-4)


@pytest.mark.parametrize("lp", [torch.bfloat16, torch.float16])
def test_tiled64_fold_epilogue_equals_projection_of_the_stored_output(cuda, lp):
    """gvf_attn_tiled64_fold_fwd + gvf_attn_fold_reduce = (the 16-bit rows gvf_attn_tiled64_fwd stores) @ W^T + bias in fp32: the fold multiplies
    the SAME rounded values, only the order of the fp32 additions differs (per head, then over the heads)."""
    from gvfdiffusion_amd.ops import dit_ops
    B, T, Lq, Lk, H, n_out = 2, 3, 2048 + 77, 300, 3, 14
    C = H * 64
    g = torch.Generator().manual_seed(9)
    q = torch.randn((B, Lq, C), generator=g).to(lp).to(cuda)
    kv = torch.randn((B * T * Lk, 2 * C), generator=g).to(lp).to(cuda)
    w = (torch.randn((n_out, C), generator=g) * C ** -0.5).to(lp).to(cuda)
    bias = torch.randn(n_out, generator=g).to(cuda)
    kt, vt = dit_ops.attention_pack_kv64(kv, B * T, Lk, H, 0, C)
    o16 = torch.empty((B, T, Lq, C), dtype=lp, device=cuda)
    dit_ops.attention_tiled64(q, kt, vt, o16, B, T, Lq, Lk, H, (Lq * C, 0, C), (T * Lq * C, Lq * C, C), T, 1)
    ref = o16.float() @ w.float().t() + bias
    frags = dit_ops.attention_fold_pack(w, n_out, H)
    part = torch.full((B * T, H, Lq, 16), float("nan"), dtype=torch.float32, device=cuda)
    dit_ops.attention_tiled64_fold(q, kt, vt, frags, part, B, T, Lq, Lk, H, (Lq * C, 0, C), T, 1)
    assert torch.isfinite(part).all() and float(part[..., n_out:].abs().max()) == 0.0        # rows past n_out of the padded matrix are zeros
    big = torch.full((B, T, Lq + 5, n_out), float("nan"), device=cuda)                       # a view with its own row / set strides
    dit_ops.attention_fold_reduce(part, bias, big[:, :, 5:], B * T, H, Lq, n_out, (Lq + 5) * n_out, n_out)
    assert torch.isnan(big[:, :, :5]).all()
    got = big[:, :, 5:]
    assert float((got - ref).abs().max()) < 2e-5 * float(ref.abs().max()), float((got - ref).abs().max())


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_decode_is_the_same_with_and_without_the_folded_epilogue(cuda, dtype):
    import os
    from gvfdiffusion_amd.model.autoencoder import GSKLTemporalVariationalAutoEncoder
    torch.manual_seed(4)
    m = GSKLTemporalVariationalAutoEncoder(depth=2, dim=384, queries_dim=384, output_dim=14, num_inputs=512, num_latents=128, latent_dim=16, heads=6,
                                           dim_head=64, num_timesteps=3)
    _randomise(m)
    m = m.to(cuda).set_compute_dtype(dtype)
    m.max_chunk_rows = 3 * 2048                     # three chunks, the last one ragged
    x = torch.randn(3, 128, 16, device=cuda)
    qs = torch.randn(1, 5000, 14, device=cuda)
    y1 = m.decode(x, qs)
    os.environ["GVF_VAE_FOLD"] = "0"
    try:
        y0 = m.decode(x, qs)
    finally:
        del os.environ["GVF_VAE_FOLD"]
    assert float((y1 - y0).abs().max()) < 2e-5 * float(y0.abs().max())


@pytest.mark.parametrize("lp", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,F,K", [(300, 96, 64), (12288, 3072, 768), (64, 32, 128)])
def test_geglu_epilogue_equals_projection_then_geglu(cuda, lp, M, F, K):
    """EPI_GEGLU_16 on the row-interleaved projection == the store epilogue followed by gvf_geglu, bit for bit: both round value and gate to the
    operand type before the erf GELU and the product (model/autoencoder.py:90-93 under autocast)."""
    from gvfdiffusion_amd.ops import dit_ops, vae_ops
    g = torch.Generator().manual_seed(M + F)
    a = torch.randn((M, K), generator=g).to(lp).to(cuda)
    w = (torch.randn((2 * F, K), generator=g) / K ** 0.5)
    b = torch.randn(2 * F, generator=g)
    hid = torch.empty((M, 2 * F), dtype=lp, device=cuda)
    dit_ops.gemm(a, w.to(lp).to(cuda), b.to(cuda), hid, dit_ops.EPI_STORE_BF16)
    ref = vae_ops.geglu_bf16(hid)
    wi, bi = dit_ops.geglu_interleave(w, b)
    out = torch.full((M, F + 8), float("nan"), dtype=lp, device=cuda)
    dit_ops.gemm(a, wi.to(lp).to(cuda), bi.to(cuda), out[:, :F], dit_ops.EPI_GEGLU_16)
    assert torch.isnan(out[:, F:]).all()
    assert torch.equal(out[:, :F], ref)
    with pytest.raises(Exception):                                # 2F must be a multiple of 64
        dit_ops.gemm(a, wi[:32].to(lp).to(cuda), None, out[:, :16], dit_ops.EPI_GEGLU_16)
