"""oracle/vae_ref.py (torch restatement of the motion-VAE decode) against the reference's own
model/autoencoder.py output (tests/golden/vae_small_golden.npz)."""
import json
import os

import numpy as np
import torch

from oracle import vae_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_decode_matches_reference():
    g = np.load(os.path.join(GOLD, "vae_small_golden.npz"))
    cfg = json.loads(bytes(g["cfg_json"]).decode())
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    with torch.no_grad():
        y = vae_ref.vae_decode(sd, cfg, torch.from_numpy(g["x"]), torch.from_numpy(g["queries"]), cfg["num_timesteps"])
    assert y.shape == g["y"].shape
    assert np.abs(y.numpy() - g["y"]).max() < 1e-6, np.abs(y.numpy() - g["y"]).max()   # measured: 0.0 (same torch ops)
    with torch.no_grad():
        yb = vae_ref.vae_decode(sd, cfg, torch.from_numpy(g["x"]), torch.from_numpy(g["queries"]), cfg["num_timesteps"], "bf16")
    assert float((yb - y).norm() / y.norm()) < 3e-2


def test_manifest_is_the_reference_layout():
    man = json.load(open(os.path.join(GOLD, "vae_manifest.json")))
    assert len(man["state_dict"]) == 121 and man["config"]["heads"] == 12 and man["config"]["dim"] == 768
    assert man["state_dict"]["decoder_cross_attn.fn.to_kv.weight"] == [1536, 768]
    assert man["state_dict"]["layers.11.1.fn.net.0.weight"] == [6144, 768]
