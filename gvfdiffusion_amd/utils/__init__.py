"""Host-side drivers that mirror the reference's utils/ package for the hot path."""
from .inference_utils import orbit_cameras, render_sample_frames, render_and_save_images, seed_everything, align_gaussian_to_canonical  # noqa: F401
from .points import fps, sample_gs, pad_static_gs, get_gaussian_tensor  # noqa: F401
from .image_ops import resize_pad_crop_u8, resample_table  # noqa: F401
from .in_flight import run_in_flight, streams_for  # noqa: F401
