"""CPU: the oracle's static-VAE backbone (oracle/sparse_vae_ref.py) against tests/golden/sparse_vae_golden.npz -- outputs
of the reference's own SparseTransformerVAE.encode / decode (model/sparse_voxel_diffusion/sparse_transformer_vae.py)
run in the build container (tests/golden/make_golden.py::gen_sparse_vae), both qkv channel layouts."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import sparse_vae_ref as ref

GOLD = os.path.join(os.path.dirname(__file__), "golden", "sparse_vae_golden.npz")


def load():
    z = np.load(GOLD)
    cfg = json.loads(bytes(z["cfg_json"]).decode())
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    return z, cfg, sd


@pytest.mark.parametrize("tag,old", [("new", False), ("old", True)])
def test_encode_decode_fp32_matches_reference(tag, old):
    z, cfg, sd = load()
    cfg = dict(cfg, use_old_attn_impl=old)
    feats, coords = torch.from_numpy(z["feats"]), torch.from_numpy(z["coords"])
    mean, logvar = ref.encode(sd, cfg, feats, coords)
    assert np.abs(mean.numpy() - z[f"{tag}_mean"]).max() < 2e-5
    assert np.abs(logvar.numpy() - z[f"{tag}_logvar"]).max() < 2e-5
    out = ref.decode(sd, cfg, torch.from_numpy(z[f"{tag}_mean"]), coords)
    assert np.abs(out.numpy() - z[f"{tag}_out"]).max() < 2e-5


def test_bf16_restatement_is_close():
    z, cfg, sd = load()
    feats, coords = torch.from_numpy(z["feats"]), torch.from_numpy(z["coords"])
    mean, _ = ref.encode(sd, cfg, feats, coords, "bf16")
    rel = float(np.linalg.norm(mean.numpy() - z["new_mean"]) / np.linalg.norm(z["new_mean"]))
    assert rel < 2e-2, rel


def test_window_ids_group_tokens_like_the_device_partition():
    """The oracle's grouping and the product's calc_window_partition (device-side sort; runs on CPU tensors too) agree."""
    from types import SimpleNamespace
    from gvfdiffusion_amd.sparse.attention.windowed_attn import calc_window_partition
    z, _, _ = load()
    coords = torch.from_numpy(z["coords"])
    for shift in (0, 4):
        gid = ref.window_ids(coords, 8, shift)
        fwd, bwd, lens, _ = calc_window_partition(SimpleNamespace(coords=coords), 8, shift)
        assert torch.equal(fwd[bwd], torch.arange(coords.shape[0]))
        start = 0
        for n in lens:
            assert len(set(gid[fwd[start:start + n]].tolist())) == 1
            start += n
        assert len(lens) == len(torch.unique(gid))


SLAT = os.path.join(os.path.dirname(__file__), "golden", "slat_decoder_golden.npz")


def slat_state_dict(z, tag):
    sd = {k[7:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_rms.")}
    sd.update({k[len(tag) + 4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"sd_{tag}.")})
    return {k: v for k, v in sd.items() if tag == "rms" or "rms_norm" not in k}


@pytest.mark.parametrize("tag,rms", [("rms", True), ("plain", False)])
def test_slat_decoder_rows_match_reference(tag, rms):
    """TRELLIS SLatGaussianDecoder (tests/golden/make_golden.py::gen_slat_decoder), with and without QK-RMSNorm."""
    z = np.load(SLAT)
    cfg = dict(json.loads(bytes(z["cfg_json"]).decode()), qk_rms_norm=rms)
    sd = slat_state_dict(z, tag)
    rows = ref.slat_decode_rows(sd, cfg, torch.from_numpy(z["feats"]), torch.from_numpy(z["coords"]))
    assert np.abs(rows.numpy() - z[f"{tag}_rows"]).max() < 2e-5
