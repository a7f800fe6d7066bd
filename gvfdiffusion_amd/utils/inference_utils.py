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
import torch.nn.functional as F

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


def azimuth_cameras(azimuths, elevation: float = 0.0, radius: float = 2.0) -> torch.Tensor:
    """World-to-camera matrices for arbitrary azimuths (degrees), built as `orbit_cameras` builds its ring
    (utils/inference_utils.py:53-58)."""
    convert = np.array([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float64)
    out = []
    for azi in azimuths:
        pose = convert @ _orbit_pose_opengl(elevation, float(azi), radius)
        pose[:3, 1:3] *= -1
        out.append(np.linalg.inv(pose))
    return torch.from_numpy(np.stack(out)).float()


def build_rotation(r: torch.Tensor) -> torch.Tensor:
    """Quaternion (r, x, y, z), any norm -> rotation matrix (utils/script_util.py:102-123)."""
    q = r / r.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(dim=1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)


def matrix_to_quaternion(m: torch.Tensor) -> torch.Tensor:
    """Rotation matrices (N, 3, 3) -> unit quaternions (r, x, y, z) with r >= 0 (pytorch3d.transforms.matrix_to_quaternion,
    a third-party call at utils/inference_utils.py:174, restated from its definition: of the four candidate solutions the
    one with the largest divisor is taken)."""
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.reshape(-1, 9).unbind(dim=1)
    q_abs = torch.sqrt(torch.clamp(torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22,
                                                1 - m00 - m11 + m22], dim=1), min=0.0))
    cand = torch.stack([torch.stack([q_abs[:, 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=1),
                        torch.stack([m21 - m12, q_abs[:, 1] ** 2, m10 + m01, m02 + m20], dim=1),
                        torch.stack([m02 - m20, m10 + m01, q_abs[:, 2] ** 2, m12 + m21], dim=1),
                        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[:, 3] ** 2], dim=1)], dim=1)
    cand = cand / (2.0 * q_abs[:, :, None].clamp(min=0.1))
    best = cand[torch.arange(m.shape[0], device=m.device), q_abs.argmax(dim=1)]
    return torch.where(best[:, :1] < 0, -best, best)


def _mask_extent(mask: torch.Tensor) -> torch.Tensor:
    """(F, H, W) bool -> (F,) max(height, width) of the mask's bounding box as the reference measures it
    (`max - min` of the indices, :71-77); -1 where the mask is empty."""
    out = []
    for axis in (2, 1):                                   # rows that contain a pixel, then columns
        hit = mask.any(dim=axis)
        n = hit.shape[1]
        idx = torch.arange(n, device=mask.device)
        lo = torch.where(hit, idx, n).amin(dim=1)
        hi = torch.where(hit, idx, -1).amax(dim=1)
        out.append(hi - lo)
    ext = torch.maximum(out[0], out[1])
    return torch.where(mask.flatten(1).any(dim=1), ext, torch.full_like(ext, -1))


@torch.no_grad()
def align_gaussian_to_canonical(static_gs_model, canonical_image, canonical_alpha, intrinsics, static_vae, id, device,
                                in_the_wild=True, clip_score=None, chunk_frames: int = 60):
    """Find the azimuth (and the image-space scale) at which the static Gaussians look like the canonical view, then turn
    the model so that this view becomes the front view (utils/inference_utils.py:38-178; same arguments and return value
    `(static_gs_model, best_scale_factor)`).

    The reference renders the 360 (or 4) azimuths one at a time and scores each with an L1 term plus 0.2 x (1 - CLIP
    similarity) from a downloaded ViT-B/32.  Here the azimuths are rendered in batches by the alpha-output rasteriser
    variant, the bounding boxes come from the alpha masks on the device, and the score's CLIP term is a callback:
    `clip_score(image (3,512,512) in [0,1], canonical_image) -> similarity`; with None (no CLIP weights on this machine) the
    score is the L1 term alone."""
    renderer = static_vae.renderers["MipGS"]
    renderer.pipe.use_mip_gaussian = False                                       # :50
    azimuths = np.arange(-180, 180, 1) if in_the_wild else np.arange(-180, 180, 90)
    cams = azimuth_cameras(azimuths).to(device)
    canonical_image = canonical_image.to(device)
    cmask = (canonical_alpha.to(device) > 0.5)
    canonical_size = int(_mask_extent(cmask[None])[0])
    best = (1e8, 0, 1.0)
    if canonical_size >= 0:
        K = intrinsics.to(device)
        for s0 in range(0, len(azimuths), chunk_frames):
            out = renderer.render_frames(static_gs_model, cams[s0:s0 + chunk_frames], K, want_alpha_depth=True)
            sizes = _mask_extent(out.alpha > 0.5).tolist()
            for k, rendered_size in enumerate(sizes):
                if rendered_size < 0:
                    continue                                                      # nothing visible from here (:66-67)
                scale_factor = canonical_size / rendered_size if rendered_size > 0 else float("inf")
                if not math.isfinite(scale_factor):
                    continue
                target = int(512 * scale_factor)
                if target < 1 or target > 8192:
                    continue
                image = F.interpolate(out.rgb[k].clamp(0.0, 1.0)[None], size=(target, target), mode="bicubic", align_corners=False)[0]
                if target < 512:
                    p = max(0, (512 - target) // 2)
                    image = F.pad(image, (p, 512 - target - p, p, 512 - target - p), mode="constant", value=1.0)
                else:
                    o = (target - 512) // 2
                    image = image[:, o:o + 512, o:o + 512]
                image = image.clamp(0.0, 1.0)
                diff = float((image - canonical_image).abs().mean())
                if clip_score is not None:
                    diff += (1.0 - float(clip_score(image, canonical_image))) * 0.2
                if diff < best[0]:
                    best = (diff, int(azimuths[s0 + k]), float(scale_factor))
    _, best_azi, best_scale_factor = best
    print(f"\nID: {id} \tBest azimuth: {best_azi} \tBest scale factor: {best_scale_factor}")
    a = math.radians(-best_azi)
    rot = torch.tensor([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]], device=device, dtype=torch.float32)
    static_gs_model.from_xyz((rot @ static_gs_model.get_xyz.T).T)
    static_gs_model.from_rotation(matrix_to_quaternion(rot @ build_rotation(static_gs_model.get_rotation).to(device)))
    return static_gs_model, best_scale_factor


def frame_schedule(n_timesteps: int, n_views: int) -> List[Tuple[int, int]]:
    """(timestep, camera) of every frame in the reference's nesting order (`for t in range(32): for cam in
    range(128)`, inference_utils.py:257-259)."""
    return [(t, c) for t in range(n_timesteps) for c in range(n_views)]


_STREAM_POOL = {}


def _side_streams(dev, n: int):
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), n)
    if key not in _STREAM_POOL:
        _STREAM_POOL[key] = [torch.cuda.Stream(device=dev) for _ in range(n)]
    return _STREAM_POOL[key]


@torch.no_grad()
def render_sample_frames(renderer, gaussian, pred_delta: torch.Tensor, intrinsics: torch.Tensor,
                         extrinsics: Optional[torch.Tensor] = None, n_views: int = 128,
                         timesteps: Optional[Sequence[int]] = None, n_valid: Optional[int] = None,
                         chunk_frames: int = 96, as_uint8: bool = True, resize_to: Optional[int] = None,
                         out_size: int = 512, streams: int = 2) -> Iterator[Tuple[List[Tuple[int, int]], torch.Tensor]]:
    """Render every (timestep, camera) view of ONE sample; yields `(schedule_chunk, frames)` with frames
    `(F, 3, H, W)` on the device, uint8 (`as_uint8`) or fp32.

    renderer: a gvfdiffusion_amd GaussianRenderer (`static_vae.renderers["MipGS"]`); gaussian: its GaussianModel
    (`static_gs_model[b]`); pred_delta: (T, P, 14) = `pred_delta[b]`; n_valid: `valid_idx[b]` (the reference slices
    `pred_delta[b][t, :valid_idx[b]]`; rows past it must belong to padding and are ignored by passing a model of
    n_valid Gaussians); extrinsics: (V, 4, 4) world-to-camera, default the 128-view orbit; resize_to: LANCZOS-resize the
    uint8 frames to resize_to x resize_to and pad (white) / centre-crop to out_size x out_size on the device
    (`target_size = int(512 * scale_factors[b])`, inference_utils.py:272-296).

    streams: chunks are independent, so `streams` of them are kept in flight on as many HIP streams (each with its own workspace):
    the HBM-bound front of one chunk (projection, binning) runs under the VALU-bound compositing of the other -- measured +10 % frames/s
    at 24 x 800x800 frames of 262 144 Gaussians per chunk (`scripts/rast_two_streams.py`).  The first chunk is rendered synchronously
    (it sizes the workspace); later chunks are enqueued without a host sync and their instance count is checked when they are
    handed out (a chunk that overflowed its workspace is rendered again, synchronously).  Frames are identical for any `streams`;
    the yielded tensors are safe to use on the caller's current stream."""
    if resize_to is not None and not as_uint8:
        raise ValueError("resize_to works on the uint8 frames")
    dev = pred_delta.device
    T = pred_delta.shape[0]
    if n_valid is not None and n_valid != pred_delta.shape[1]:
        pred_delta = pred_delta[:, :n_valid].contiguous()
    ext = (orbit_cameras(n_views) if extrinsics is None else extrinsics).to(dev)
    K = intrinsics.to(dev)
    sched = [(t, c) for t in (range(T) if timesteps is None else timesteps) for c in range(ext.shape[0])]
    chunks = [sched[s0:s0 + chunk_frames] for s0 in range(0, len(sched), chunk_frames)]
    old_mip = renderer.pipe.use_mip_gaussian
    renderer.pipe.use_mip_gaussian = True                 # inference_utils.py:231

    blocks = renderer.make_frames(ext, K)                 # the cameras' blocks once per sample (cached across samples on the same orbit)

    def render(part, cap=None):
        # (as_uint8: the frames leave the compositing kernel as uint8 -- round 6; GVF_RENDER_FUSED_U8=0: the fp32 frames + frames_to_uint8, a
        # measurement switch, same bits)
        fused = as_uint8 and os.environ.get("GVF_RENDER_FUSED_U8", "1") != "0"
        out = renderer.render_frames(gaussian, None, None, delta_pc=pred_delta, frames=renderer.frames_with_delta_index(blocks, part),
                                     max_rendered=cap, sync=cap is None, as_uint8=fused)
        frames = frames_to_uint8(out.rgb) if (as_uint8 and not fused) else out.rgb
        if resize_to is not None:
            frames = resize_pad_crop_u8(frames, resize_to, out_size=out_size, pad_value=255)
        return frames, out.num_rendered

    try:
        if streams <= 1 or len(chunks) <= 2:
            for part in chunks:
                yield part, render(part)[0]
            return
        pool = _side_streams(dev, streams)
        frames, nr = render(chunks[0])                    # synchronous: measures the instance count

        def counts(nr_):                                  # per-frame instance counts (device uint32 viewed as int32) as int64
            return nr_.to(torch.int64).bitwise_and(0xFFFFFFFF)
        per_frame = int(counts(nr).max()) + 1
        yield chunks[0], frames
        inflight = []

        def hand_out():
            part, fr, nr_, ev, cap = inflight.pop(0)
            ev.synchronize()
            if int(counts(nr_).sum()) > cap:              # rare: a denser view than the estimate
                return part, render(part)[0]
            cur = torch.cuda.current_stream(dev)          # the consumer's stream NOW (it may have changed between next() calls)
            cur.wait_event(ev)
            fr.record_stream(cur)
            return part, fr

        for i, part in enumerate(chunks[1:]):
            s = pool[i % streams]
            s.wait_stream(torch.cuda.current_stream(dev))   # the sample's tensors were produced on the caller's stream
            cap = int(per_frame * len(part) * 1.5) + 4096
            with torch.cuda.stream(s):
                fr, nr_ = render(part, cap)
                ev = torch.cuda.Event()
                ev.record(s)
            inflight.append((part, fr, nr_, ev, cap))
            if len(inflight) >= streams:
                yield hand_out()
        while inflight:
            yield hand_out()
    finally:
        # a consumer that stops early (break / exception) leaves chunks queued on the side streams that still read pred_delta, the
        # Gaussians and the camera tensors -- all allocated on the caller's stream: order the caller's stream behind them, so that
        # the caching allocator cannot hand that memory out while they run
        if dev.type == "cuda" and streams > 1:
            cur = torch.cuda.current_stream(dev)
            for s in _side_streams(dev, streams):
                cur.wait_stream(s)
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
    written, pending = [], []

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
                    pending.append(pool.submit(save, host[k], path))
                    written.append(path)
        for f in pending:
            f.result()                                     # a failed write raises here instead of disappearing
    return written
