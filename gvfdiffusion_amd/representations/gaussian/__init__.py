from .gaussian_model import GaussianModel, Gaussian, build_rotation, build_scaling_rotation, strip_symmetric

__all__ = ["GaussianModel", "Gaussian", "build_rotation", "build_scaling_rotation", "strip_symmetric"]
