"""The torch DiT oracle (oracle/dit_ref.py) against outputs of the reference's model/dit.py
(tests/golden/dit_small_golden.npz: reduced width, full structure, weights inside the fixture;
dit_full_golden.npz: configs/diffusion.yml at B=1,T=24 with seed-generated weights and inputs)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import dit_ref
from gvfdiffusion_amd import synthetic

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_small():
    g = np.load(os.path.join(GOLD, "dit_small_golden.npz"))
    cfg = json.loads(bytes(g["cfg_json"]).decode())
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    return g, cfg, sd


def test_subops_match_reference():
    g, cfg, sd = load_small()
    t = torch.from_numpy(g["t"])
    assert np.abs(dit_ref.timestep_embedding(t).numpy() - g["t_freq"]).max() < 1e-6      # cos first, then sin
    ape = dit_ref.absolute_position_embedding(torch.from_numpy(g["xyz"]), cfg["model_channels"])
    assert np.abs(ape.numpy() - g["ape"]).max() < 1e-6
    assert (ape[..., (cfg["model_channels"] // 3 // 2) * 6:] == 0).all()                  # zero padding tail
    gamma = sd["blocks.0.spatial_self_attn.q_rms_norm.gamma"]
    rms = dit_ref.rms_norm_heads(torch.from_numpy(g["rms_in"]), gamma, "fp32")
    assert np.abs(rms.numpy() - g["rms_out"]).max() < 1e-5


def test_small_model_forward_matches_reference():
    g, cfg, sd = load_small()
    args = [torch.from_numpy(g[k]) for k in ("x", "t", "cond_images", "static_latent", "xyz")]
    with torch.no_grad():
        y, inter = dit_ref.dit_forward(sd, cfg, *args, precision="fp32", return_intermediates=True)
    assert np.abs(inter["t_emb"].numpy() - g["t_emb"]).max() < 1e-5
    assert np.abs(inter["h0"].numpy() - g["h0"]).max() < 1e-5
    assert np.abs(inter["block0"].numpy() - g["block0"]).max() < 5e-5
    err = np.abs(y.numpy() - g["y"]).max()
    assert err < 5e-5, err
    # the bf16-emulating mode stays close to fp32 (sanity of the rounding model, not a parity claim)
    with torch.no_grad():
        yb = dit_ref.dit_forward(sd, cfg, *args, precision="bf16")
    rel = float((yb - y).norm() / y.norm())
    assert rel < 3e-2, rel
    # ... and is more accurate than the reference's OWN bf16 autocast run of model/dit.py on the same inputs
    # (tests/golden/dit_autocast_golden.npz, make_golden.py::gen_dit_autocast).  With the small projections in fp32 (dit_ref.FP32_SITES,
    # the HIP pipeline's placement) this two-block model loses almost all of its bf16 error: 2.4e-4 against the reference's 4.2e-3, and
    # less than the reference's fp16 autocast (5.2e-4); with every site in bf16 the same oracle gives 3.1e-3 (not optimistic either)
    ac = np.load(os.path.join(GOLD, "dit_autocast_golden.npz"))
    ref_bf16, ref_fp16 = float(ac["small_rel_l2_bf16"]), float(ac["small_rel_l2_fp16"])
    assert abs(float((torch.from_numpy(ac["small_y_bf16"]) - torch.from_numpy(g["y"])).norm() / torch.from_numpy(g["y"]).norm()) - ref_bf16) < 1e-6
    assert 0.02 * ref_bf16 < rel <= ref_fp16, (rel, ref_bf16, ref_fp16)
    saved = dit_ref.FP32_SITES
    try:
        dit_ref.FP32_SITES = ()
        with torch.no_grad():
            rel_all = float((dit_ref.dit_forward(sd, cfg, *args, precision="bf16") - y).norm() / y.norm())
    finally:
        dit_ref.FP32_SITES = saved
    assert 0.3 * ref_bf16 < rel_all <= 1.1 * ref_bf16, (rel_all, ref_bf16)


def test_small_model_without_temporal_attention_matches_reference():
    """no_temporal_attn=True (model/dit.py:241-260, 358): the reference's own forward of that variant, make_golden.py::gen_dit_notemporal."""
    import json
    g = np.load(os.path.join(GOLD, "dit_small_notemporal_golden.npz"))
    cfg = json.loads(bytes(g["cfg_json"]).decode())
    assert cfg["no_temporal_attn"] is True
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    assert not any("temporal" in k and "weight" in k and v.numel() > 0 for k, v in sd.items() if "adaLN_modulation_temporal" in k)
    args = [torch.from_numpy(g[k]) for k in ("x", "t", "cond_images", "static_latent", "xyz")]
    with torch.no_grad():
        y = dit_ref.dit_forward(sd, cfg, *args, precision="fp32")
    err = np.abs(y.numpy() - g["y"]).max()
    assert err < 5e-5, err


def test_small_model_with_one_head_of_64_matches_reference():
    """num_heads = 1 at 64 channels, i.e. head_dim 64 (the reference takes any num_heads, model/dit.py:337): the reference's own forward of that
    variant (make_golden.py::gen_dit_hd64) pins the oracle's head handling (split, MultiHeadRMSNorm gains [H][d], scale d^-0.5)."""
    import json
    g = np.load(os.path.join(GOLD, "dit_small_hd64_golden.npz"))
    cfg = json.loads(bytes(g["cfg_json"]).decode())
    assert cfg["num_heads"] == 1 and cfg["model_channels"] == 64
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    args = [torch.from_numpy(g[k]) for k in ("x", "t", "cond_images", "static_latent", "xyz")]
    with torch.no_grad():
        y = dit_ref.dit_forward(sd, cfg, *args, precision="fp32")
    err = np.abs(y.numpy() - g["y"]).max()
    assert err < 5e-5, err


def test_full_config_forward_matches_reference_two_frames():
    """configs/diffusion.yml at full width (512 channels, 16 heads, 12 blocks, 1370 image and 4096 static tokens) on T = 2 frames: 0.55 TFLOP,
    cheap on any host.  Fixture: the reference's own fp32 forward (tests/golden/make_golden.py dit_full_t2)."""
    man = json.load(open(os.path.join(GOLD, "dit_manifest.json")))
    sd = synthetic.dit_state_dict(man["state_dict"], seed=0)
    inp = synthetic.dit_inputs(B=1, T=2, seed=1)
    gold = np.load(os.path.join(GOLD, "dit_full_t2_golden.npz"))
    with torch.no_grad():
        y = dit_ref.dit_forward(sd, man["config"], inp["x"], inp["t"], inp["cond_images"], inp["static_latent"],
                                inp["deformation_position_xyz"], precision="fp32")
    err = np.abs(y.numpy() - gold["y"])
    assert err.max() < 2e-3 and err.mean() < 5e-5, (err.max(), err.mean())               # fp32 accumulation-order noise


@pytest.mark.skipif((os.cpu_count() or 1) < 32 and os.environ.get("GVF_FULL_ORACLE") != "1",
                    reason="5.04 TFLOP of fp32 on the host: ~9 min on 8 cores (under a minute on the GPU box's 128); the two-frame test above holds the "
                           "same widths and context lengths, the device tests hold this fixture at T = 24; GVF_FULL_ORACLE=1 forces it")
def test_full_config_forward_matches_reference():
    """configs/diffusion.yml, B=1, T=24, 1370 image tokens, 4096 static tokens (5.04 TFLOP on the CPU)."""
    man = json.load(open(os.path.join(GOLD, "dit_manifest.json")))
    assert len(man["state_dict"]) == 446
    sd = synthetic.dit_state_dict(man["state_dict"], seed=0)
    inp = synthetic.dit_inputs(B=1, T=24, seed=1)
    gold = np.load(os.path.join(GOLD, "dit_full_golden.npz"))
    with torch.no_grad():
        y = dit_ref.dit_forward(sd, man["config"], inp["x"], inp["t"], inp["cond_images"], inp["static_latent"],
                                inp["deformation_position_xyz"], precision="fp32")
    err = np.abs(y.numpy() - gold["y"])
    assert err.max() < 2e-3 and err.mean() < 5e-5, (err.max(), err.mean())               # fp32 accumulation-order noise
