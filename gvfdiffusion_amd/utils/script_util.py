"""The two names the inference path imports from the reference's utils/script_util.py: `create_gaussian_diffusion`
(inference_dpm_latent.py:27; schedule tables for NoiseScheduleVP) and `build_rotation` (utils/inference_utils.py:15, used by
align_gaussian_to_canonical).  The model / training factories of that file are out of scope."""
from ..model.gaussian_diffusion import create_gaussian_diffusion  # noqa: F401
from .inference_utils import build_rotation  # noqa: F401

__all__ = ["create_gaussian_diffusion", "build_rotation"]
