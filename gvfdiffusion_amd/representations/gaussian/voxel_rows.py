"""Per-voxel network output rows -> GaussianModel: the conversion both decoders of the reference end with
(model/sparse_voxel_diffusion/sparse_vae.py:114-182 `SparseVAE.to_representation`, trellis/models/structured_latent_vae/
decoder_gs.py:78-115 `SLatGaussianDecoder.to_representation`).  A row holds, for `n` Gaussians of one active voxel, the channel
groups [_xyz 3n | _features_dc 3n | _scaling 3n | _rotation 4n | _opacity n]; every group is scaled by its `lr` factor, and the
position group becomes an offset from the voxel centre, squashed by tanh to a fraction of the voxel pitch."""
from typing import Dict, Optional

import torch

from .gaussian_model import GaussianModel

GROUPS = (("_xyz", (3,)), ("_features_dc", (1, 3)), ("_scaling", (3,)), ("_rotation", (4,)), ("_opacity", (1,)))


def gaussian_row_layout(n: int, start: int = 0) -> Dict[str, dict]:
    """{group: {shape: (n, ...), size, range: (first channel, one past the last)}} in row order, from channel `start`."""
    out = {}
    for name, tail in GROUPS:
        size = n
        for t in tail:
            size *= t
        out[name] = {"shape": (n, *tail), "size": size, "range": (start, start + size)}
        start += size
    return out


def rows_to_gaussian(rows: torch.Tensor, voxel_xyz: torch.Tensor, resolution: int, layout: Dict[str, dict], lr: Dict[str, float],
                     offset_scale: float, perturbation: Optional[torch.Tensor], model_kwargs: dict) -> GaussianModel:
    """rows (L, channels) of one sample, voxel_xyz (L, 3) integer voxel coordinates -> a GaussianModel holding L * n
    Gaussians.  offset_scale: the offset is tanh(.) / resolution * offset_scale (1 = stay inside the voxel)."""
    rep = GaussianModel(sh_degree=0, aabb=[-0.5, -0.5, -0.5, 1.0, 1.0, 1.0], device=rows.device, **model_kwargs)
    centre = (voxel_xyz.float() + 0.5) / resolution
    for name, spec in layout.items():
        lo, hi = spec["range"]
        v = rows[:, lo:hi].reshape(-1, *spec["shape"]) * lr[name]
        if name == "_xyz":
            if perturbation is not None:
                v = v + perturbation
            v = centre.unsqueeze(1) + torch.tanh(v) / resolution * offset_scale
        setattr(rep, name, v.flatten(0, 1))
    return rep
