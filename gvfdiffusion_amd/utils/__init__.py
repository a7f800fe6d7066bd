"""Host-side drivers that mirror the reference's utils/ package for the hot path."""
from .inference_utils import orbit_cameras, render_sample_frames  # noqa: F401
from .points import fps, sample_gs, pad_static_gs, get_gaussian_tensor  # noqa: F401
