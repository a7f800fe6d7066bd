"""Host-side drivers that mirror the reference's utils/ package for the hot path."""
from .inference_utils import orbit_cameras, render_sample_frames  # noqa: F401
