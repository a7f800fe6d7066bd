"""CPU: the oracle's motion-VAE encode (oracle/vae_ref.py::vae_encode, delta_interp; oracle/points_ref.py FPS) against
tests/golden/vae_encode_golden.npz -- outputs of the reference's own GSKLTemporalVariationalAutoEncoder.encode
(model/autoencoder.py:502-550) run in the build container with torch_cluster.fps / pytorch3d knn_points replaced by the
deterministic stand-ins described in tests/golden/make_golden.py::gen_vae_encode."""
import json
import os

import numpy as np
import torch

from oracle import points_ref, vae_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden", "vae_encode_golden.npz")


def load():
    z = np.load(GOLD)
    cfg = json.loads(bytes(z["cfg_json"]).decode())
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    return z, cfg, sd


def sampled_rows(z, cfg):
    gs = [z["gs0"], z["gs1"]]
    L = cfg["num_latents"]
    ptr = [0, gs[0].shape[0], gs[0].shape[0] + gs[1].shape[0]]
    idx = points_ref.fps_indices(np.concatenate([g[:, :3] for g in gs]), ptr, [L, L], [0, 0])
    return np.concatenate(gs)[idx].reshape(2, L, 14)


def test_fps_selects_the_fixture_rows():
    z, cfg, _ = load()
    assert np.array_equal(sampled_rows(z, cfg), z["sampled"])


def test_delta_interp_matches_reference():
    z, cfg, _ = load()
    static_pc, delta_pc = torch.from_numpy(z["static_pc"]), torch.from_numpy(z["delta_pc"])
    est = vae_ref.delta_interp(torch.from_numpy(z["sampled"][..., :3]), static_pc, delta_pc + static_pc[:, None],
                               int(z["knn_k"]), float(z["beta"]))
    assert np.abs(est.numpy() - z["est"]).max() < 1e-6


def test_encode_fp32_matches_reference():
    z, cfg, sd = load()
    mean, logvar, _ = vae_ref.vae_encode(sd, cfg, torch.from_numpy(z["static_pc"]), torch.from_numpy(z["delta_pc"]),
                                         torch.from_numpy(z["sampled"][..., :3]), int(z["knn_k"]), float(z["beta"]))
    assert np.abs(mean.numpy() - z["mean"]).max() < 2e-5
    assert np.abs(logvar.numpy() - z["logvar"]).max() < 2e-5
    # DiagonalGaussianDistribution.kl (:303-340): 0.5 * mean over (1, 2) of mean^2 + var - 1 - logvar
    kl = 0.5 * (mean ** 2 + logvar.exp() - 1.0 - logvar).mean(dim=(1, 2))
    assert np.abs(kl.numpy() - z["kl"]).max() < 1e-5


def test_encode_bf16_restatement_is_close():
    z, cfg, sd = load()
    args = (sd, cfg, torch.from_numpy(z["static_pc"]), torch.from_numpy(z["delta_pc"]), torch.from_numpy(z["sampled"][..., :3]),
            int(z["knn_k"]), float(z["beta"]))
    mean, _, _ = vae_ref.vae_encode(*args, precision="bf16")
    rel = float(np.linalg.norm(mean.numpy() - z["mean"]) / np.linalg.norm(z["mean"]))
    assert rel < 2e-2, rel
