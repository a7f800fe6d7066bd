from .gaussian_render import GaussianRenderer, render, intrinsics_to_projection

__all__ = ["GaussianRenderer", "render", "intrinsics_to_projection"]
