"""gvfdiffusion_amd.utils.inference_utils.align_gaussian_to_canonical against the REFERENCE's function
(utils/inference_utils.py:37-177) run in the build container on the same stand-in renderer (tests/align_util.py):
tests/golden/align_golden.npz by tests/golden/make_golden.py::gen_align, CLIP term neutralised on both sides.
Pins the bounding-box / scale-factor / bicubic resize / pad-crop / L1 / arg-min / rotation-update logic on the CPU."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(__file__))
import align_util

GOLD = os.path.join(os.path.dirname(__file__), "golden", "align_golden.npz")


class _Renderer:
    """Stand-in for GaussianRenderer: the batched entry point the MI355X driver calls, fed by the same synthetic views."""

    def __init__(self, n_views):
        self.pipe = SimpleNamespace(use_mip_gaussian=True)
        self.n_views, self.next = n_views, 0

    def render_frames(self, model, extrinsics, intrinsics, want_alpha_depth=False, **kw):
        views = [align_util.view(self.next + k, self.n_views) for k in range(extrinsics.shape[0])]
        self.next += extrinsics.shape[0]
        return SimpleNamespace(rgb=torch.stack([v[0] for v in views]), alpha=torch.stack([v[1] for v in views]))


def test_alignment_matches_reference_fixture(capsys):
    from gvfdiffusion_amd.utils.inference_utils import align_gaussian_to_canonical
    g = np.load(GOLD)
    for tag, wild in (("wild", True), ("coarse", False)):
        v_star, zoom = int(g[f"{tag}.params"][0]), float(g[f"{tag}.params"][1])
        n = 360 if wild else 4
        rend = _Renderer(n)
        model = align_util.ToyGaussians(seed=3)
        canon_rgb, canon_alpha = align_util.canonical(v_star, n, zoom)
        vae = SimpleNamespace(renderers={"MipGS": rend})
        model, scale = align_gaussian_to_canonical(model, canon_rgb, canon_alpha, torch.eye(3), vae, 0, torch.device("cpu"),
                                                   in_the_wild=wild, chunk_frames=50)
        printed = capsys.readouterr().out
        assert f"Best azimuth: {int(g[f'{tag}.best_azimuth'])} " in printed.replace("\t", " "), printed
        assert rend.pipe.use_mip_gaussian is False                                   # :50 side effect on the caller's renderer
        assert abs(scale - float(g[f"{tag}.scale_factor"])) < 1e-6 * float(g[f"{tag}.scale_factor"])
        assert np.abs(model.get_xyz.numpy() - g[f"{tag}.xyz"]).max() < 1e-6
        q, q_ref = model.get_rotation.numpy(), g[f"{tag}.rotation"]
        assert np.minimum(np.abs(q - q_ref).max(1), np.abs(q + q_ref).max(1)).max() < 1e-5     # a quaternion and its negative are one rotation
