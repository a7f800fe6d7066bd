"""Drop-in for the `diff_gaussian_rasterization` package of the mip-splatting fork, as imported by
the reference at renderers/gaussian_render.py:106 (settings :110-125, call :198-206): same class
names, same NamedTuple fields (incl. kernel_size, subpixel_offset), returns (color, radii)."""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib, rasterizer as _r


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    kernel_size: float
    subpixel_offset: torch.Tensor
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizer(nn.Module):
    _MODE = _lib.RAST_MODE_MIP

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def _run(self, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, want_ad, means2D=None):
        rs = self.raster_settings
        st = _r.make_settings(rs.image_height, rs.image_width, rs.sh_degree, self._MODE,
                              getattr(rs, "kernel_size", 0.0), rs.scale_modifier, rs.bg, rs.prefiltered, rs.debug)
        fr = _r.make_frame(rs.viewmatrix, rs.projmatrix, rs.campos, rs.tanfovx, rs.tanfovy)
        sub = getattr(rs, "subpixel_offset", None)
        if sub is not None and not bool(torch.any(sub != 0)):
            sub = None  # the reference always passes zeros (gaussian_render.py:108)
        return _r.rasterize(st, fr, means3D, opacities, shs=shs, colors_precomp=colors_precomp, scales=scales,
                            rotations=rotations, cov3D_precomp=cov3D_precomp, subpixel_offset=sub,
                            want_alpha_depth=want_ad, means2D=means2D)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        # means2D only carries screen-space gradients upstream (screenspace_points.grad); the forward pass ignores it.
        out = self._run(means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, False, means2D)
        return out["color"], out["radii"]
