"""renderers/gaussian_render_all_delta.py of the reference -- the module BASELINE.json's north_star names.  Upstream it is
the rgb-only subset of renderers/gaussian_render.py (every Gaussian attribute takes its per-frame delta, the mip rasteriser
returns `(color, radii)`, no depth / alpha in the result dict; nothing in the reference imports it).  Here it is the same
facade over the same HIP rasteriser with the result reduced to that module's keys."""
from ..attrdict import edict
from . import gaussian_render as _g
from .gaussian_render import intrinsics_to_projection  # noqa: F401

__all__ = ["render", "GaussianRenderer", "intrinsics_to_projection"]


def render(viewpoint_camera, pc, pipe, bg_color, delta_pc=None, detach_static=False, scaling_modifier=1.0, override_color=None):
    out = _g.render(viewpoint_camera, pc, pipe, bg_color, delta_pc=delta_pc, detach_static=detach_static,
                    scaling_modifier=scaling_modifier, override_color=override_color)
    return edict({k: v for k, v in out.items() if k not in ("depth", "alpha")})


class GaussianRenderer(_g.GaussianRenderer):
    def render(self, gausssian, extrinsics, intrinsics, delta_pc=None, detach_static=False, colors_overwrite=None, patch_mask=None):
        ret = super().render(gausssian, extrinsics, intrinsics, delta_pc=delta_pc, detach_static=detach_static,
                             colors_overwrite=colors_overwrite, patch_mask=patch_mask)
        return edict({"rgb": ret["rgb"]})
