"""gvfdiffusion_amd -- MI355X-native (gfx950) implementation of GVFDiffusion's 4D
render-and-denoise hot path behind the reference's own operator surface.

  diff_gaussian_rasterization / diff_gauss   GaussianRasterizationSettings + GaussianRasterizer
  renderers.GaussianRenderer                 renderers/gaussian_render.py of the reference
  representations.gaussian.GaussianModel     representations/gaussian/gaussian_model.py
  model.dit.DiT, model.attention             model/dit.py, model/attention/*
  model.dpmsolver                            NoiseScheduleVP / model_wrapper / DPM_Solver
Compute lives in hand-written HIP kernels (csrc/) behind the C ABI declared in include/*.h.
"""
__version__ = "0.1.0"
