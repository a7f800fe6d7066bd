"""Drop-in for the `diff_gauss` package (slothfulxtx/diff-gaussian-rasterization) as imported by the
reference at renderers/gaussian_render.py:127 (settings :128-141, call :208-220): no mip filter
(+0.3 px^2 dilation), returns (color, depth, normal, alpha, radii, extra) -- the caller branches on
len(ret) (:217-220) and consumes color, depth, alpha, radii only."""
from typing import NamedTuple

import torch

from . import _lib
from .diff_gaussian_rasterization import GaussianRasterizer as _MipRasterizer


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizer(_MipRasterizer):
    _MODE = _lib.RAST_MODE_DILATE

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, extra_attrs=None):
        out = self._run(means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, True, means2D)
        H, W = out["alpha"].shape
        # depth = sum_i z_i a_i T_i, alpha = 1 - T_final, both (1,H,W) as upstream; the normal map and
        # the extra-attribute channel are never read by the reference (gaussian_render.py:220-238).
        normal = torch.zeros((3, H, W), dtype=torch.float32, device=out["color"].device)
        return out["color"], out["depth"][None], normal, out["alpha"][None], out["radii"], None
