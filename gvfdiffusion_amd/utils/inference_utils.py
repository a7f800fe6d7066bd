"""Batched render driver -- the device-resident counterpart of `render_and_save_images`
(utils/inference_utils.py:208-297 of the reference; SURVEY.md section 8a row C1, section 8f NEXT #2).

The reference renders B x 32 x 128 views one at a time, each followed by `.cpu()`, a PIL LANCZOS resize and a PNG
write.  Here all (timestep, camera) pairs of a sample go through `GaussianRenderer.render_frames` in chunks (one
fused launch sequence per chunk, deltas applied inside the preprocess kernel) and leave the device as uint8
(`gvf_rgb_to_u8` = the reference's clamp(0,1) * 255 -> astype('uint8'), inference_utils.py:276-281), optionally already
resized / padded / cropped to 512x512 on the device with Pillow's own arithmetic (`resize_to=`, csrc/resize.hip, bit-identical
to :283-296).  `render_and_save_images` keeps the reference's signature and file names; PNG encoding is the only host work
left (a writer thread per call), MP4 assembly needs imageio and is skipped when it is absent.
"""
import math
from typing import Iterator, List, Optional, Sequence, Tuple

import os
import random
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from ..rasterizer import frames_to_uint8
from .image_ops import resize_pad_crop_u8


def _orbit_pose_opengl(elevation_deg: float, azimuth_deg: float, radius: float) -> np.ndarray:
    """Camera-to-world of an OpenGL camera (x right, y up, z backward) orbiting the origin of a y-up world.
    Restates `kiui.cam.orbit_camera(elevation, azimuth, radius, opengl=True)`, which the reference imports from the
    `kiui` package (not in the reference tree, unpinned): eye = r (cos e sin a, -sin e, cos e cos a), look-at origin."""
    e, a = math.radians(elevation_deg), math.radians(azimuth_deg)
    eye = np.array([radius * math.cos(e) * math.sin(a), -radius * math.sin(e), radius * math.cos(e) * math.cos(a)])
    fwd = eye / np.linalg.norm(eye)                       # OpenGL: the camera looks along -z, so +z points to the eye
    right = np.cross(np.array([0.0, 1.0, 0.0]), fwd)
    right /= np.linalg.norm(right)
    up = np.cross(fwd, right)
    pose = np.eye(4)
    pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, up, fwd, eye
    return pose


def orbit_cameras(n_views: int = 128, elevation: float = 0.0, radius: float = 2.0) -> torch.Tensor:
    """World-to-camera matrices (n_views, 4, 4) of the reference's inference orbit
    (utils/inference_utils.py:239-254): azimuths arange(0, 360, 360 / n_views); orbit pose -> y-up to z-up
    (`convert_mat`) -> OpenGL to COLMAP axes (flip y, z) -> inverse."""
    convert = np.array([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float64)
    out = []
    for azi in np.arange(0, 360, 360 / n_views):
        pose = convert @ _orbit_pose_opengl(elevation, float(azi), radius)
        pose[:3, 1:3] *= -1
        out.append(np.linalg.inv(pose))
    return torch.from_numpy(np.stack(out)).float()


def frame_schedule(n_timesteps: int, n_views: int) -> List[Tuple[int, int]]:
    """(timestep, camera) of every frame in the reference's nesting order (`for t in range(32): for cam in
    range(128)`, inference_utils.py:257-259)."""
    return [(t, c) for t in range(n_timesteps) for c in range(n_views)]


@torch.no_grad()
def render_sample_frames(renderer, gaussian, pred_delta: torch.Tensor, intrinsics: torch.Tensor,
                         extrinsics: Optional[torch.Tensor] = None, n_views: int = 128,
                         timesteps: Optional[Sequence[int]] = None, n_valid: Optional[int] = None,
                         chunk_frames: int = 96, as_uint8: bool = True, resize_to: Optional[int] = None,
                         out_size: int = 512) -> Iterator[Tuple[List[Tuple[int, int]], torch.Tensor]]:
    """Render every (timestep, camera) view of ONE sample; yields `(schedule_chunk, frames)` with frames
    `(F, 3, H, W)` on the device, uint8 (`as_uint8`) or fp32.

    renderer: a gvfdiffusion_amd GaussianRenderer (`static_vae.renderers["MipGS"]`); gaussian: its GaussianModel
    (`static_gs_model[b]`); pred_delta: (T, P, 14) = `pred_delta[b]`; n_valid: `valid_idx[b]` (the reference slices
    `pred_delta[b][t, :valid_idx[b]]`; rows past it must belong to padding and are ignored by passing a model of
    n_valid Gaussians); extrinsics: (V, 4, 4) world-to-camera, default the 128-view orbit; resize_to: LANCZOS-resize the
    uint8 frames to resize_to x resize_to and pad (white) / centre-crop to out_size x out_size on the device
    (`target_size = int(512 * scale_factors[b])`, inference_utils.py:272-296)."""
    if resize_to is not None and not as_uint8:
        raise ValueError("resize_to works on the uint8 frames")
    dev = pred_delta.device
    T = pred_delta.shape[0]
    if n_valid is not None and n_valid != pred_delta.shape[1]:
        pred_delta = pred_delta[:, :n_valid].contiguous()
    ext = (orbit_cameras(n_views) if extrinsics is None else extrinsics).to(dev)
    sched = [(t, c) for t in (range(T) if timesteps is None else timesteps) for c in range(ext.shape[0])]
    old_mip = renderer.pipe.use_mip_gaussian
    renderer.pipe.use_mip_gaussian = True                 # inference_utils.py:231
    try:
        for s0 in range(0, len(sched), chunk_frames):
            part = sched[s0:s0 + chunk_frames]
            e = ext[torch.tensor([c for _, c in part], device=dev)]
            out = renderer.render_frames(gaussian, e, intrinsics.to(dev), delta_pc=pred_delta,
                                         delta_index=[t for t, _ in part])
            frames = frames_to_uint8(out.rgb) if as_uint8 else out.rgb
            if resize_to is not None:
                frames = resize_pad_crop_u8(frames, resize_to, out_size=out_size, pad_value=255)
            yield part, frames
    finally:
        renderer.pipe.use_mip_gaussian = old_mip


def seed_everything(seed: int):
    """utils/inference_utils.py:200-205."""
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def render_and_save_images(args, static_vae, static_gs_model, pred_delta, model_kwargs, valid_idx, img_id, accelerator,
                           scale_factors, save_dir="inference_images", out_root: Optional[str] = None, n_timesteps: int = 32,
                           n_views: int = 128, chunk_frames: int = 96):
    """Same arguments, same files as the reference (utils/inference_utils.py:208-306):
    `<root>/<save_dir>/rank_RR_render_IIIIII_cam_CCC_timesteps_TT.png`, 512x512, for every (timestep < 32, camera < 128) of
    every sample.  Frames are rendered, quantised, resized and padded / cropped on the device; the host only encodes PNGs.
    `out_root` replaces `logger.get_dir()` (the reference's logger is not part of this package; default: args.exp_name).
    `accelerator` needs `.device` and `.process_index` only.  Returns the list of files written."""
    from PIL import Image
    root = out_root if out_root is not None else getattr(args, "exp_name", ".")
    s_path = os.path.join(root, save_dir)
    os.makedirs(s_path, exist_ok=True)
    os.makedirs(os.path.join(root, "inference_videos"), exist_ok=True)
    renderer = static_vae.renderers["MipGS"]
    dev = accelerator.device
    intrinsics = model_kwargs["cams"]["intrinsics"][0][0].to(dev)
    rank = getattr(accelerator, "process_index", 0)
    written = []

    def save(arr, path):
        Image.fromarray(arr).save(path)

    with ThreadPoolExecutor(max_workers=8) as pool:
        for b in range(pred_delta.shape[0]):
            T = min(n_timesteps, pred_delta.shape[1])
            target = int(512 * scale_factors[b])
            for part, frames in render_sample_frames(renderer, static_gs_model[b], pred_delta[b][:T].to(dev), intrinsics, n_views=n_views,
                                                     n_valid=valid_idx[b], chunk_frames=chunk_frames, resize_to=target, out_size=512):
                host = frames.permute(0, 2, 3, 1).contiguous().cpu().numpy()          # one transfer per chunk
                for k, (t, c) in enumerate(part):
                    path = os.path.join(s_path, f"rank_{rank:02d}_render_{img_id + b:06d}_cam_{c:03d}_timesteps_{t:02d}.png")
                    pool.submit(save, host[k], path)
                    written.append(path)
    return written
