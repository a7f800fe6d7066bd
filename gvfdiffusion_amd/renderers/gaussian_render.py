"""Renderer facade with the reference's surface (renderers/gaussian_render.py:57-369; the unimported
renderers/gaussian_render_all_delta.py is its rgb-only, mip-only subset).

  intrinsics_to_projection(intrinsics, near, far)                       :57-82
  render(camera, pc, pipe, bg, delta_pc, detach_static, scaling_modifier, override_color)  :85-238
  GaussianRenderer(rendering_options).render(gaussian, extrinsics, intrinsics, delta_pc, ...) :242-369
`pipe` / `rendering_options` stay writable attribute dicts because callers mutate them
(inference_dpm_latent.py:161-162, utils/inference_utils.py:50,238).  The rasteriser classes come from
this package's HIP-backed drop-ins instead of the external CUDA wheels.  `render_frames()` is the
MI355X addition: all (frame, camera) pairs of a sample in ONE launch sequence with the Gaussian
delta activations fused into the rasteriser preprocess (utils/inference_utils.py:256-269 loop).
"""
import math
import threading as _threading

import numpy as np
import torch
import torch.nn.functional as F

from ..attrdict import edict
from .. import rasterizer as _r
from .. import _lib


def intrinsics_to_projection(intrinsics: torch.Tensor, near: float, far: float) -> torch.Tensor:
    """Normalised OpenCV intrinsics (fx,fy,cx,cy in [0,1] units) -> GL-style 4x4 perspective."""
    fx, fy, cx, cy = intrinsics[0, 0], intrinsics[1, 1], intrinsics[0, 2], intrinsics[1, 2]
    proj = torch.zeros((4, 4), dtype=intrinsics.dtype, device=intrinsics.device)
    proj[0, 0] = 2 * fx
    proj[1, 1] = 2 * fy
    proj[0, 2] = 2 * cx - 1
    proj[1, 2] = -2 * cy + 1
    proj[2, 2] = far / (far - near)
    proj[2, 3] = near * far / (near - far)
    proj[3, 2] = 1.0
    return proj


def _camera(extrinsics, intrinsics, near, far, size):
    view = extrinsics
    persp = intrinsics_to_projection(intrinsics, near, far)
    return edict({
        "image_height": size, "image_width": size,
        "FoVx": 2 * torch.atan(0.5 / intrinsics[0, 0]), "FoVy": 2 * torch.atan(0.5 / intrinsics[1, 1]),
        "znear": near, "zfar": far,
        "world_view_transform": view.T.contiguous(),
        "projection_matrix": persp.T.contiguous(),
        "full_proj_transform": (persp @ view).T.contiguous(),
        "camera_center": torch.inverse(view)[:3, 3],
    })


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, delta_pc=None, detach_static=False,
           scaling_modifier=1.0, override_color=None):
    """One frame through the rasteriser operator; returns the reference's result dict."""
    tanfovx = math.tan(float(viewpoint_camera.FoVx) * 0.5)
    tanfovy = math.tan(float(viewpoint_camera.FoVy) * 0.5)
    H, W = int(viewpoint_camera.image_height), int(viewpoint_camera.image_width)
    dev = pc.get_xyz.device
    # zero tensor whose .grad receives the screen-space (2D mean) gradients, as at gaussian_render.py:96-100
    screenspace_points = torch.zeros_like(pc.get_xyz)
    if torch.is_grad_enabled():
        screenspace_points = torch.zeros_like(pc.get_xyz, requires_grad=True) + 0
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass

    common = dict(image_height=H, image_width=W, tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color,
                  scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
                  projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
                  campos=viewpoint_camera.camera_center, prefiltered=False, debug=pipe.debug)
    if pipe.use_mip_gaussian:
        from ..diff_gaussian_rasterization import GaussianRasterizer, GaussianRasterizationSettings
        settings = GaussianRasterizationSettings(kernel_size=pipe.kernel_size, subpixel_offset=None, **common)
    else:
        from ..diff_gauss import GaussianRasterizer, GaussianRasterizationSettings
        settings = GaussianRasterizationSettings(**common)
    rasterizer = GaussianRasterizer(raster_settings=settings)

    shs = None
    opacity = pc.get_opacity
    if delta_pc is not None:
        means3D = pc.get_xyz_with_delta(delta_pc[..., :3], detach=detach_static)
        scales = pc.get_scaling_with_delta(delta_pc[..., 3:6], detach=detach_static)
        rotations = pc.get_rotation_with_delta(delta_pc[..., 6:10], detach=detach_static)
        if delta_pc.shape[1] > 10:
            shs = pc.get_features_with_delta(delta_pc[..., 10:13].unsqueeze(1), detach=detach_static)
            opacity = pc.get_opacity_with_delta(delta_pc[..., 13:], detach=detach_static)
    else:
        means3D, scales, rotations = pc.get_xyz, pc.get_scaling, pc.get_rotation
    cov3D_precomp = pc.get_covariance(scaling_modifier) if pipe.compute_cov3D_python else None
    if cov3D_precomp is not None:
        scales = rotations = None

    colors_precomp = None
    if override_color is not None:
        colors_precomp, shs = override_color, None
    elif pipe.convert_SHs_python:
        from .sh_utils import eval_sh
        feats = pc.get_features if shs is None else shs
        shs_view = feats.transpose(1, 2).reshape(-1, 3, (pc.max_sh_degree + 1) ** 2)
        dirs = F.normalize(means3D - viewpoint_camera.camera_center[None], dim=1)
        colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dirs) + 0.5, 0.0)
        shs = None
    elif shs is None:
        shs = pc.get_features

    ret = rasterizer(means3D=means3D, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp,
                     opacities=opacity, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    depth = alpha = None
    if len(ret) == 2:
        image, radii = ret
    else:
        image, depth, _normal, alpha, radii, _extra = ret
    return edict({"render": image, "depth": depth, "alpha": alpha.squeeze() if alpha is not None else None,
                  "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii})


class GaussianRenderer:
    """Same constructor / attributes / render() contract as the reference class (:242-369)."""

    def __init__(self, rendering_options={}) -> None:
        self.pipe = edict({"use_mip_gaussian": False, "kernel_size": 0.1, "convert_SHs_python": False,
                           "compute_cov3D_python": False, "scale_modifier": 1.0, "debug": False})
        self.rendering_options = edict({"resolution": None, "near": None, "far": None, "ssaa": 1,
                                        "bg_color": "random"})
        self.rendering_options.update(rendering_options)
        self.bg_color = None

    def _background(self, device):
        if self.rendering_options["bg_color"] == "random":
            bg = torch.zeros(3, dtype=torch.float32, device=device)
            if np.random.rand() < 0.5:
                bg += 1
        else:
            bg = torch.tensor(self.rendering_options["bg_color"], dtype=torch.float32, device=device)
        self.bg_color = bg
        return bg

    def render(self, gausssian, extrinsics, intrinsics, delta_pc=None, detach_static=False, colors_overwrite=None,
               patch_mask=None):
        opts = self.rendering_options
        size = opts["resolution"] * opts["ssaa"]
        bg = self._background(extrinsics.device)
        cam = _camera(extrinsics, intrinsics, opts["near"], opts["far"], size)
        out = render(cam, gausssian, self.pipe, bg, delta_pc=delta_pc, detach_static=detach_static,
                     override_color=colors_overwrite, scaling_modifier=self.pipe.scale_modifier)
        if opts["ssaa"] > 1:
            out.render = F.interpolate(out.render[None], size=(opts["resolution"],) * 2, mode="bicubic",
                                       align_corners=False, antialias=True).squeeze()
        ret = edict({"rgb": out["render"]})
        if out.get("depth") is not None:
            ret["depth"] = out["depth"]
        if out.get("alpha") is not None:
            ret["alpha"] = out["alpha"]
        return ret

    # ---- MI355X batched path -----------------------------------------------------------------
    def make_frames(self, extrinsics, intrinsics, delta_index=None):
        """Camera blocks (GvfRastFrame) of F views, built exactly as render() builds its camera dict."""
        opts = self.rendering_options
        size = int(opts["resolution"]) * int(opts["ssaa"])
        Fn = extrinsics.shape[0]
        # The render loops call this with the SAME camera tensors for every sample / chunk (the fixed orbit of
        # inference_dpm_latent.py:262-269): building 24 blocks costs ~2 ms of host time (one device read, then an inverse, a matmul
        # and three .tolist() per frame) against ~1.5 ms of GPU work for the frames themselves.  Keep the last few sets, keyed (1) on
        # the tensors' identity and version (the entry holds the tensors, so an address cannot be recycled while it is cached) -- a hit
        # costs nothing -- and (2) on the CONTENT of the host copies: the chunked driver (utils/inference_utils.render_sample_frames)
        # slices a fresh `ext[idx]` per chunk, so identity never repeats there, but every sample walks the same orbit.  One renderer may
        # be shared by the threads of utils/in_flight.py: the list is read and written under a lock.
        di = None if delta_index is None else tuple(int(d) for d in delta_index)
        common = (float(opts["near"]), float(opts["far"]), size, di)
        key = (extrinsics.data_ptr(), extrinsics._version, tuple(extrinsics.shape), intrinsics.data_ptr(), intrinsics._version,
               tuple(intrinsics.shape)) + common
        lock = self.__dict__.setdefault("_frame_cache_lock", _threading.Lock())
        cache = self.__dict__.setdefault("_frame_cache", [])
        with lock:
            for k, _, _, frames in cache:
                if k == key:
                    return frames
        held = (extrinsics, intrinsics)
        if intrinsics.dim() == 2:
            intrinsics = intrinsics[None].expand(Fn, 3, 3)
        ext_c, int_c = extrinsics.detach().float().cpu(), intrinsics.detach().float().cpu()
        ckey = (ext_c.numpy().tobytes(), int_c.contiguous().numpy().tobytes()) + common
        with lock:
            for _, ck, _, frames in cache:
                if ck == ckey:
                    return frames
        frames = []
        for f in range(Fn):
            cam = _camera(ext_c[f], int_c[f], opts["near"], opts["far"], size)
            frames.append(_r.make_frame(cam.world_view_transform, cam.full_proj_transform, cam.camera_center,
                                        math.tan(float(cam.FoVx) * 0.5), math.tan(float(cam.FoVy) * 0.5),
                                        -1 if di is None else di[f]))
        with lock:
            cache.append((key, ckey, held, frames))
            del cache[:-8]
        return frames

    @staticmethod
    def frames_with_delta_index(blocks, pairs):
        """Camera blocks for a schedule: `pairs` = (delta_index, camera) per frame, `blocks` = make_frames(...) of the cameras.  A block is
        copied (160 bytes) and only its delta_index set: no camera arithmetic, no device read.  The chunked driver builds the orbit's
        blocks ONCE per sample and takes every chunk's frames from them (the per-chunk `make_frames` of a fresh `ext[idx]` cost a device
        read and ~80 us of host arithmetic per frame: ~8 ms per 96-frame chunk against 4.8 ms of GPU work)."""
        out = []
        for t, c in pairs:
            fr = _lib.GvfRastFrame.from_buffer_copy(blocks[c])
            fr.delta_index = int(t)
            out.append(fr)
        return out

    def render_frames(self, gaussian, extrinsics, intrinsics, delta_pc=None, delta_index=None,
                      want_alpha_depth=False, max_rendered=None, sync=True, frames=None, as_uint8=False):
        """Render F frames of one sample in a single fused launch sequence.

        extrinsics (F,4,4) world-to-camera; intrinsics (3,3) or (F,3,3) normalised; delta_pc (T,P,14)
        or None; delta_index: F ints selecting the delta slice per frame (default: frame f -> min(f,T-1),
        -1 = static).  frames: ready camera blocks (`make_frames` / `frames_with_delta_index`) instead of
        extrinsics / intrinsics / delta_index.  Returns edict(rgb (F,3,H,W) [, alpha, depth (F,H,W)], num_rendered (F,)).
        as_uint8: rgb comes back as uint8 = the reference's frame post-process (clamp(0,1) * 255 -> uint8, utils/inference_utils.py:280-286)
        done in the compositing kernel's epilogue -- bit-identical to rasterizer.frames_to_uint8(rgb), without the fp32 frames' trip through
        HBM.  Falls back to that two-step form where the fused one does not apply (ssaa > 1, alpha / depth wanted, the dilation mode)."""
        opts = self.rendering_options
        ssaa = int(opts["ssaa"])
        size = int(opts["resolution"]) * ssaa          # supersampled render, down-sampled below as render() does (gaussian_render.py:355-360)
        dev = extrinsics.device if frames is None else gaussian._xyz.device
        bg = self._background(dev)
        T = 0 if delta_pc is None else (1 if delta_pc.dim() == 2 else delta_pc.shape[0])
        if frames is None:
            if delta_index is None:
                delta_index = [min(f, T - 1) if T > 0 else -1 for f in range(extrinsics.shape[0])]
            frames = self.make_frames(extrinsics, intrinsics, delta_index)
        elif any(fr.delta_index >= T for fr in frames):
            raise ValueError("a camera block selects a delta slice that delta_pc does not have")
        mode = _lib.RAST_MODE_MIP if self.pipe.use_mip_gaussian else _lib.RAST_MODE_DILATE
        st = _r.make_settings(size, size, gaussian.active_sh_degree, mode, self.pipe.kernel_size,
                              self.pipe.scale_modifier, bg, False, self.pipe.debug)
        want_ad = want_alpha_depth or not self.pipe.use_mip_gaussian
        fused_u8 = as_uint8 and ssaa == 1 and not want_ad
        out = _r.rasterize_batched(st, frames, gaussian.activation_struct(), gaussian._xyz, gaussian.get_features,
                                   gaussian._scaling, gaussian._rotation, gaussian._opacity, delta=delta_pc,
                                   want_alpha_depth=want_ad, max_rendered=max_rendered, sync=sync, color_u8=fused_u8)
        rgb = out["color"]
        if ssaa > 1:                                   # all F frames in one bicubic antialiased resize
            rgb = F.interpolate(rgb, size=(int(opts["resolution"]),) * 2, mode="bicubic", align_corners=False, antialias=True)
        if as_uint8 and not fused_u8:
            rgb = _r.frames_to_uint8(rgb)
        ret = edict({"rgb": rgb, "num_rendered": out["num_rendered"]})
        if out["alpha"] is not None:                   # (depth / alpha stay at the supersampled size, as in render())
            ret["alpha"], ret["depth"] = out["alpha"], out["depth"]
        return ret
