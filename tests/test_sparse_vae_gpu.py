"""Static-VAE backbone on the HIP kernels (gvfdiffusion_amd/model/sparse_voxel_diffusion) against the torch oracle
(oracle/sparse_vae_ref.py, pinned to the reference by tests/test_oracle_sparse_vae.py).

Tolerances (relative L2, measured value + ~50 %, per operand type): vs the same-type oracle (same rounding points; differences =
accumulation order, exp2 softmax) and vs the fp32 oracle."""
import json
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL16, TOL32 = 5e-3, 7.5e-3          # bf16 operands: measured 1.4e-3 .. 3.3e-3 vs the bf16-placement oracle, 3.6e-3 .. 4.8e-3 vs fp32
# fp16 operands (use_fp16=True / convert_to_fp16() / set_compute_dtype("fp16") -- the reference's torso type): measured 2.5e-4 .. 4.7e-4 vs the
# fp16-placement oracle, 4.2e-4 .. 6.0e-4 vs fp32
TOL16_FP16, TOL32_FP16 = 7e-4, 9e-4
GOLD = os.path.join(os.path.dirname(__file__), "golden", "sparse_vae_golden.npz")


def _rel(a, b):
    return float((a - b).norm() / b.norm())


def _golden():
    z = np.load(GOLD)
    cfg = json.loads(bytes(z["cfg_json"]).decode())
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    return z, cfg, sd


def _voxels(res, counts, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for b, n in enumerate(counts):
        c = torch.unique(torch.randint(0, res, (n * 3, 3), generator=g), dim=0)
        c = c[torch.randperm(c.shape[0], generator=g)[:n]]
        out.append(torch.cat([torch.full((c.shape[0], 1), b), c], dim=1))
    return torch.cat(out).int()


def _run(cfg, sd, feats, coords, z_in=None, dtype=None):
    from gvfdiffusion_amd import sparse as sp
    from gvfdiffusion_amd.model.sparse_voxel_diffusion import SparseTransformerVAE
    from oracle import sparse_vae_ref as ref
    m = SparseTransformerVAE(**cfg)
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    if dtype is not None:
        m.set_compute_dtype(dtype)
    lp = "fp16" if m._lp() == torch.float16 else "bf16"        # the operand type this pass contracts (module default: use_fp16 -> fp16)
    tol16, tol32 = (TOL16_FP16, TOL32_FP16) if lp == "fp16" else (TOL16, TOL32)
    x = sp.SparseTensor(feats.cuda(), coords.cuda())
    z, mean, logvar = m.encode(x, sample_posterior=False, return_raw=True)
    assert torch.equal(z.feats, mean) and torch.equal(z.coords, x.coords)
    from oracle_cache import oracle_cached
    # (oracle results cached on disk by content: the fp32 passes serve both operand types and the child process of
    # test_kv_resident_attention_variant_forced_everywhere)
    r32 = oracle_cached("svae_encode", ref, (cfg, sd, feats, coords, "fp32"), lambda: ref.encode(sd, cfg, feats, coords))
    r16 = oracle_cached("svae_encode", ref, (cfg, sd, feats, coords, lp), lambda: ref.encode(sd, cfg, feats, coords, lp))
    for got, a, b, name in ((mean.cpu(), r16[0], r32[0], "mean"), (logvar.cpu(), r16[1], r32[1], "logvar")):
        print(f"static vae {name} [{lp}]: rel-L2 vs {lp} oracle {_rel(got, a):.2e}, vs fp32 oracle {_rel(got, b):.2e}")
        assert _rel(got, a) < tol16 and _rel(got, b) < tol32
    zin = r32[0] if z_in is None else z_in
    y = m.decode(sp.SparseTensor(zin.cuda(), coords.cuda()))
    d32 = oracle_cached("svae_decode", ref, (cfg, sd, zin, coords, "fp32"), lambda: ref.decode(sd, cfg, zin, coords))
    d16 = oracle_cached("svae_decode", ref, (cfg, sd, zin, coords, lp), lambda: ref.decode(sd, cfg, zin, coords, lp))
    print(f"static vae decode [{lp}]: rel-L2 vs {lp} oracle {_rel(y.feats.cpu(), d16):.2e}, vs fp32 oracle {_rel(y.feats.cpu(), d32):.2e}")
    assert _rel(y.feats.cpu(), d16) < tol16 and _rel(y.feats.cpu(), d32) < tol32
    return m, x, y


@pytest.mark.parametrize("dtype", [None, "fp16", "bf16"])
@pytest.mark.parametrize("old", [False, True])
def test_golden_config_both_qkv_layouts(cuda, old, dtype):
    z, cfg, sd = _golden()
    cfg = dict(cfg, use_old_attn_impl=old)
    m, x, y = _run(cfg, sd, torch.from_numpy(z["feats"]), torch.from_numpy(z["coords"]), dtype=dtype)
    tag = "old" if old else "new"
    # and against the reference's own outputs (fp32): the fixture's mean through the device decode
    assert _rel(y.feats.cpu(), torch.from_numpy(z[f"{tag}_out"])) < 5e-2


def _random_sd(m, seed):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in m.state_dict().items():
        sd[k] = torch.randn(v.shape, generator=g) * (1.0 / math.sqrt(v.shape[1]) if v.dim() == 2 else 0.1)
    return sd


def test_released_width_ragged_batch(cuda):
    """768 channels, 12 heads of 64, window 8 on a 64^3 grid (configs/diffusion.yml: static_vae), 4 of the 12 blocks,
    two samples of different size -- hundreds of windows of 1..~40 tokens."""
    from gvfdiffusion_amd.model.sparse_voxel_diffusion import SparseTransformerVAE
    cfg = dict(resolution=64, in_channels=1024, model_channels=768, out_channels=112, latent_channels=8, num_blocks=4,
               num_heads=12, mlp_ratio=4, attn_mode="swin", window_size=8, use_fp16=True, use_old_attn_impl=False, norm_output=True)
    sd = _random_sd(SparseTransformerVAE(**cfg), 1)
    coords = _voxels(64, (3000, 1777), 2)
    feats = torch.randn((coords.shape[0], 1024), generator=torch.Generator().manual_seed(3))
    m, _, _ = _run(cfg, sd, feats, coords)                   # use_fp16=True: fp16 operands, as the reference's torso
    assert m._lp() == torch.float16
    _run(cfg, sd, feats, coords, dtype="bf16")
    m.convert_to_fp32()
    assert m._lp() == torch.bfloat16 and m.dtype == torch.float32


def test_full_attention_mode_and_single_block_module(cuda):
    from gvfdiffusion_amd import sparse as sp
    from gvfdiffusion_amd.model.sparse_voxel_diffusion import SparseTransformerBlock
    from oracle import sparse_vae_ref as ref
    torch.manual_seed(0)
    blk = SparseTransformerBlock(128, num_heads=2, attn_mode="full", modulated=False)
    sd = {"b." + k: v for k, v in _random_sd(blk, 4).items()}
    blk.load_state_dict({k[2:]: v for k, v in sd.items()})
    coords = _voxels(16, (150, 90), 5)
    feats = torch.randn((coords.shape[0], 128), generator=torch.Generator().manual_seed(6))
    y = blk.cuda()(sp.SparseTensor(feats.cuda(), coords.cuda())).feats.cpu()
    want = ref.block(feats, coords[:, 0].long(), sd, "b", 2, "bf16", False)          # full attention: groups = samples
    assert _rel(y, want) < TOL16
    with pytest.raises(NotImplementedError):
        SparseTransformerBlock(128, num_heads=2, modulated=True)


def test_framework_representation_and_render(cuda):
    """SparseVAE: decode -> GaussianModel per sample (layout ranges, lr factors, perturbation, soft_invoxel offsets) ->
    render through the rasteriser.  configs/diffusion.yml: static_vae.framework."""
    from gvfdiffusion_amd import sparse as sp
    from gvfdiffusion_amd.model.sparse_voxel_diffusion import SparseTransformerVAE, SparseVAE
    from gvfdiffusion_amd.model.sparse_voxel_diffusion.sparse_vae import hammersley_sequence
    from rast_util import camera_block
    z, cfg, sd = _golden()
    rep_cfg = {"MipGS": {"lr": {"_xyz": 1.0, "_features_dc": 1.0, "_opacity": 1.0, "_scaling": 1.0, "_rotation": 0.1},
                         "perturb_offset": True, "reg_mode": "soft_invoxel", "voxel_size": 1.5, "num_gaussians": 8,
                         "2d_filter_kernel_size": 0.1, "3d_filter_kernel_size": 0.0009, "scaling_bias": 0.004, "opacity_bias": 0.1,
                         "scaling_activation": "softplus"}}
    backbone = SparseTransformerVAE(**cfg)
    backbone.load_state_dict(sd, strict=True)
    backbone = backbone.cuda()
    vae = SparseVAE({"vae": backbone}, resolution=cfg["resolution"], representation_config=rep_cfg)
    assert vae.layouts["MipGS"]["_opacity"]["range"] == (104, 112)
    assert hammersley_sequence(3, 5, 8) == pytest.approx([5 / 8, 0.625, 2 / 3 + 1 / 9])   # 5 = 101b -> .101b; 5 = 12 (base 3) -> .21 (base 3)
    coords = torch.from_numpy(z["coords"]).cuda()
    x = sp.SparseTensor(torch.from_numpy(z["feats"]).cuda(), coords)
    reps, aux = vae.encode_decode_no_render(x, return_aux=True)
    assert len(reps["MipGS"]) == 2 and aux["mean"].shape == (coords.shape[0], 8)
    rows = aux["x"].feats[aux["x"].layout[1]]
    g = reps["MipGS"][1]
    n1 = rows.shape[0]
    assert g._xyz.shape == (n1 * 8, 3) and g._features_dc.shape == (n1 * 8, 1, 3) and g._rotation.shape == (n1 * 8, 4)
    centre = (coords[aux["x"].layout[1]][:, 1:].float() + 0.5) / 16
    off = torch.tanh(rows[:, :24].reshape(-1, 8, 3) + backbone.MipGS_perturbation) / 16 * 0.5 * 1.5
    assert torch.allclose(g._xyz, (centre[:, None] + off).reshape(-1, 3), atol=1e-6)
    assert torch.allclose(g._rotation, rows[:, 72:104].reshape(-1, 4) * 0.1, atol=1e-7)
    assert (g._xyz.reshape(-1, 8, 3) - centre[:, None]).abs().max() <= 0.75 / 16 + 1e-6      # inside 1.5 voxels
    cam = camera_block(azi=30.0, elev=15.0)
    for r in vae.renderers.values():
        r.rendering_options.resolution = 64
    with torch.no_grad():
        out = vae.render_batch(reps, cam["extrinsics"].cuda()[None].repeat(2, 1, 1), cam["intrinsics"].cuda()[None].repeat(2, 1, 1))
    rgb = out["MipGS"]["rgb"]
    assert rgb.shape == (2, 3, 64, 64) and torch.isfinite(rgb).all() and float((rgb < 0.99).float().mean()) > 0.01   # something drawn


# ---- TRELLIS SLatGaussianDecoder (trellis/models/structured_latent_vae/decoder_gs.py) --------------------------------------
SLAT = os.path.join(os.path.dirname(__file__), "golden", "slat_decoder_golden.npz")


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("tag,rms", [("rms", True), ("plain", False)])
def test_slat_gaussian_decoder(cuda, tag, rms, dtype):
    from gvfdiffusion_amd import sparse as sp
    from gvfdiffusion_amd.trellis.models import SLatGaussianDecoder
    from oracle import sparse_vae_ref as ref
    from test_oracle_sparse_vae import slat_state_dict
    z = np.load(SLAT)
    cfg = dict(json.loads(bytes(z["cfg_json"]).decode()), qk_rms_norm=rms)
    sd = slat_state_dict(z, tag)
    m = SLatGaussianDecoder(**cfg)
    m.load_state_dict(sd, strict=True)                       # incl. the offset_perturbation buffer of the reference
    assert torch.allclose(SLatGaussianDecoder(**cfg).offset_perturbation, sd["offset_perturbation"], atol=1e-6)
    m = m.cuda()
    if dtype == "fp16":
        m.convert_to_fp16()                                  # upstream's switch (base.py:93-101): the torso contracts fp16 from here on
        assert m._lp() == torch.float16 and m.dtype == torch.float16
    else:
        assert m._lp() == torch.bfloat16
    tol16, tol32 = (TOL16_FP16, TOL32_FP16) if dtype == "fp16" else (TOL16, TOL32)
    feats, coords = torch.from_numpy(z["feats"]), torch.from_numpy(z["coords"])
    x = sp.SparseTensor(feats.cuda(), coords.cuda())
    rows = m.decode_rows(x).feats.cpu()
    r16, r32 = ref.slat_decode_rows(sd, cfg, feats, coords, dtype), ref.slat_decode_rows(sd, cfg, feats, coords)
    print(f"slat decoder ({tag}, {dtype}) rows: rel-L2 vs {dtype} oracle {_rel(rows, r16):.2e}, vs fp32 oracle {_rel(rows, r32):.2e}")
    assert _rel(rows, r16) < tol16 and _rel(rows, r32) < tol32
    assert _rel(rows, torch.from_numpy(z[f"{tag}_rows"])) < tol32           # the reference's own output
    # representation: feed the reference's rows through to_representation -> its Gaussians, accessor by accessor
    reps = m.to_representation(x.replace(torch.from_numpy(z[f"{tag}_rows"]).cuda()))
    g = reps[1]
    for got, key, tol in ((g._xyz, "xyz", 1e-6), (g._rotation, "rot", 1e-7), (g.get_xyz, "get_xyz", 1e-6),
                          (g.get_scaling, "get_scaling", 1e-6), (g.get_opacity, "get_opacity", 1e-6)):
        assert np.abs(got.cpu().numpy() - z[f"{tag}_rep1_{key}"]).max() < tol, key
    assert len(m(x)) == 2


def test_empty_voxel_list(cuda):
    from gvfdiffusion_amd import sparse as sp
    from gvfdiffusion_amd.model.sparse_voxel_diffusion import SparseTransformerVAE
    z, cfg, sd = _golden()
    m = SparseTransformerVAE(**cfg)
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    x = sp.SparseTensor(torch.zeros((0, cfg["in_channels"]), device=cuda), torch.zeros((0, 4), dtype=torch.int32, device=cuda),
                        shape=torch.Size([0, cfg["in_channels"]]), layout=[])
    lat = m.encode(x, sample_posterior=False)
    assert lat.feats.shape == (0, cfg["latent_channels"])
    assert m.decode(lat).feats.shape == (0, cfg["out_channels"])
