"""Host side of the rasteriser operator: tensor checks, workspace management, C-ABI calls.

Reference seam: renderers/gaussian_render.py:110-143 (settings) and :198-220 (operator call) --
the reference imports the two classes from the external CUDA packages `diff_gaussian_rasterization`
(mip-splatting fork) and `diff_gauss`; here they are backed by libgvf_hip.so (csrc/rast.hip).
Forward only (the rasteriser backward is SURVEY.md section 8f NEXT #4).
"""
import ctypes
import math
from typing import Optional

import torch

from . import _lib

_WORKSPACES = {}  # (device index) -> uint8 tensor, grown on demand


def _workspace(device, nbytes: int) -> torch.Tensor:
    key = device.index if device.index is not None else torch.cuda.current_device()
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = None
        _WORKSPACES.pop(key, None)
        ws = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        _WORKSPACES[key] = ws
    return ws


def workspace_bytes(P: int, F: int, H: int, W: int, max_rendered: int) -> int:
    out = ctypes.c_size_t(0)
    _lib.check(_lib.lib().gvf_rast_workspace_bytes(P, F, H, W, max_rendered, ctypes.byref(out)),
               "gvf_rast_workspace_bytes")
    return int(out.value)


def _f32c(t: Optional[torch.Tensor], name: str) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# binning algorithm used when make_settings() is not told otherwise (tests flip it to cover both paths)
DEFAULT_BIN_ALGO = _lib.RAST_BIN_AUTO


def make_settings(H, W, sh_degree, mode, kernel_size, scale_modifier, bg, prefiltered=False, debug=False,
                  upstream_binning=False, bin_algo=None):
    st = _lib.GvfRastSettings()
    st.image_height, st.image_width, st.sh_degree, st.mode = int(H), int(W), int(sh_degree), int(mode)
    st.kernel_size, st.scale_modifier = float(kernel_size), float(scale_modifier)
    b = [float(x) for x in (bg.detach().cpu().tolist() if torch.is_tensor(bg) else bg)]
    st.bg[0], st.bg[1], st.bg[2] = b
    st.prefiltered, st.debug = int(bool(prefiltered)), int(bool(debug))
    st.upstream_binning = int(bool(upstream_binning))   # True: num_rendered counts upstream's 3-sigma tile rects
    st.bin_algo = int(DEFAULT_BIN_ALGO if bin_algo is None else bin_algo)   # _lib.RAST_BIN_*: see include/gvf_rast.h
    return st


def make_frame(viewmatrix, projmatrix, campos, tanfovx, tanfovy, delta_index=-1):
    """viewmatrix/projmatrix: the (4,4) tensors the reference passes (V^T and (P V)^T)."""
    fr = _lib.GvfRastFrame()
    v = viewmatrix.detach().float().reshape(-1).cpu().tolist()
    p = projmatrix.detach().float().reshape(-1).cpu().tolist()
    c = campos.detach().float().reshape(-1).cpu().tolist()
    for k in range(16):
        fr.viewmatrix[k] = v[k]
        fr.projmatrix[k] = p[k]
    for k in range(3):
        fr.campos[k] = c[k]
    fr.tanfovx, fr.tanfovy, fr.delta_index = float(tanfovx), float(tanfovy), int(delta_index)
    return fr


_CAP_HINT = {}  # (P,H,W,F) -> last capacity that sufficed


def rasterize(settings: "_lib.GvfRastSettings", frame: "_lib.GvfRastFrame", means3D, opacities, shs=None,
              colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, subpixel_offset=None,
              want_alpha_depth=False):
    """One frame, activated inputs (GaussianRasterizer.__call__).  Returns dict of device tensors."""
    _lib.require_cuda(means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp)
    if (shs is None) == (colors_precomp is None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    dev = means3D.device
    means3D = _f32c(means3D, "means3D")
    P = means3D.shape[0]
    opacities = _f32c(opacities, "opacities").reshape(-1)
    shs = _f32c(shs, "shs")
    M = 0 if shs is None else shs.shape[1]
    colors_precomp = _f32c(colors_precomp, "colors_precomp")
    scales, rotations = _f32c(scales, "scales"), _f32c(rotations, "rotations")
    cov3D_precomp = _f32c(cov3D_precomp, "cov3D_precomp")
    if subpixel_offset is not None:
        subpixel_offset = _f32c(subpixel_offset, "subpixel_offset")
    H, W = settings.image_height, settings.image_width

    color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
    alpha = torch.empty((H, W), dtype=torch.float32, device=dev) if want_alpha_depth else None
    depth = torch.empty((H, W), dtype=torch.float32, device=dev) if want_alpha_depth else None
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    nr = torch.zeros((1,), dtype=torch.int32, device=dev)

    key = (P, H, W, 1)
    cap = _CAP_HINT.get(key, max(4 * P, 1 << 16))
    while True:
        nbytes = workspace_bytes(P, 1, H, W, cap)
        ws = _workspace(dev, nbytes + 256)
        base = (ws.data_ptr() + 255) // 256 * 256
        rc = _lib.lib().gvf_rast_forward(
            ctypes.byref(settings), ctypes.byref(frame), P, M, _lib.ptr(means3D), _lib.ptr(shs),
            _lib.ptr(colors_precomp), _lib.ptr(opacities), _lib.ptr(scales), _lib.ptr(rotations),
            _lib.ptr(cov3D_precomp), _lib.ptr(subpixel_offset), ctypes.c_void_p(base), nbytes, cap,
            _lib.ptr(color), _lib.ptr(alpha), _lib.ptr(depth), _lib.ptr(radii), _lib.ptr(nr),
            _lib.current_stream(dev))
        _lib.check(rc, "gvf_rast_forward")
        n = int(nr.item()) & 0xFFFFFFFF  # upstream also syncs here (it reads the scan total to size buffers)
        if n <= cap:
            break
        cap = int(n * 1.25) + 1024
    _CAP_HINT[key] = cap
    return dict(color=color, alpha=alpha, depth=depth, radii=radii, num_rendered=n)


def rasterize_batched(settings, frames, act, xyz_raw, features_dc, scaling_raw, rotation_raw, opacity_raw,
                      delta=None, want_alpha_depth=False, want_radii=False, max_rendered=None, sync=True):
    """F frames in one call with GaussianModel activations + per-frame deltas fused in-kernel.

    frames: list of GvfRastFrame (delta_index selects the (P,14) slice of delta[n_delta,P,14]).
    With sync=False no host sync happens; the caller must check `num_rendered.sum() <= max_rendered`.
    """
    _lib.require_cuda(xyz_raw, features_dc, scaling_raw, rotation_raw, opacity_raw, delta)
    dev = xyz_raw.device
    xyz_raw, features_dc = _f32c(xyz_raw, "xyz"), _f32c(features_dc, "features_dc")
    scaling_raw, rotation_raw = _f32c(scaling_raw, "scaling"), _f32c(rotation_raw, "rotation")
    opacity_raw = _f32c(opacity_raw, "opacity").reshape(-1)
    P, M = xyz_raw.shape[0], features_dc.shape[1]
    n_delta = 0
    if delta is not None:
        delta = _f32c(delta, "delta")
        if delta.dim() == 2:
            delta = delta[None]
        assert delta.shape[1] == P and delta.shape[2] == 14, "delta must be (n,P,14) [xyz3|scale3|rot4|rgb3|op1]"
        n_delta = delta.shape[0]
    F = len(frames)
    arr = (_lib.GvfRastFrame * F)(*frames)
    H, W = settings.image_height, settings.image_width
    color = torch.empty((F, 3, H, W), dtype=torch.float32, device=dev)
    alpha = torch.empty((F, H, W), dtype=torch.float32, device=dev) if want_alpha_depth else None
    depth = torch.empty((F, H, W), dtype=torch.float32, device=dev) if want_alpha_depth else None
    radii = torch.empty((F, P), dtype=torch.int32, device=dev) if want_radii else None
    nr = torch.zeros((F,), dtype=torch.int32, device=dev)
    key = (P, H, W, F)
    cap = max_rendered if max_rendered is not None else _CAP_HINT.get(key, max(4 * P * F, 1 << 16))
    while True:
        nbytes = workspace_bytes(P, F, H, W, cap)
        ws = _workspace(dev, nbytes + 256)
        base = (ws.data_ptr() + 255) // 256 * 256
        rc = _lib.lib().gvf_rast_forward_batched(
            ctypes.byref(settings), arr, F, ctypes.byref(act), P, M, _lib.ptr(xyz_raw), _lib.ptr(features_dc),
            _lib.ptr(scaling_raw), _lib.ptr(rotation_raw), _lib.ptr(opacity_raw), _lib.ptr(delta), n_delta,
            ctypes.c_void_p(base), nbytes, cap, _lib.ptr(color), _lib.ptr(alpha), _lib.ptr(depth),
            _lib.ptr(radii), _lib.ptr(nr), _lib.current_stream(dev))
        _lib.check(rc, "gvf_rast_forward_batched")
        if not sync:
            break
        n = int(nr.to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item())
        if n <= cap:
            break
        if max_rendered is not None:
            raise _lib.GvfError(f"max_rendered={max_rendered} too small: {n} instances")
        cap = int(n * 1.25) + 1024
    if sync:
        _CAP_HINT[key] = cap
    return dict(color=color, alpha=alpha, depth=depth, radii=radii, num_rendered=nr, max_rendered=cap)


def gaussian_activate(act, xyz_raw, features_dc, scaling_raw, rotation_raw, opacity_raw, delta=None):
    """GaussianModel.get_*_with_delta on the device (csrc/rast.hip activate_kernel)."""
    _lib.require_cuda(xyz_raw)
    dev = xyz_raw.device
    xyz_raw, features_dc = _f32c(xyz_raw, "xyz"), _f32c(features_dc, "features_dc")
    scaling_raw, rotation_raw = _f32c(scaling_raw, "scaling"), _f32c(rotation_raw, "rotation")
    opacity_raw = _f32c(opacity_raw, "opacity").reshape(-1)
    delta = _f32c(delta, "delta")
    P, M = xyz_raw.shape[0], features_dc.shape[1]
    out = dict(means3D=torch.empty((P, 3), device=dev), scales=torch.empty((P, 3), device=dev),
               rotations=torch.empty((P, 4), device=dev), shs=torch.empty((P, M, 3), device=dev),
               opacities=torch.empty((P, 1), device=dev))
    rc = _lib.lib().gvf_gaussian_activate(ctypes.byref(act), P, M, _lib.ptr(xyz_raw), _lib.ptr(features_dc),
                                          _lib.ptr(scaling_raw), _lib.ptr(rotation_raw), _lib.ptr(opacity_raw),
                                          _lib.ptr(delta), _lib.ptr(out["means3D"]), _lib.ptr(out["scales"]),
                                          _lib.ptr(out["rotations"]), _lib.ptr(out["shs"]),
                                          _lib.ptr(out["opacities"]), _lib.current_stream(dev))
    _lib.check(rc, "gvf_gaussian_activate")
    return out


def sort_pairs_u64(keys: torch.Tensor, values: torch.Tensor, end_bit: int = 64):
    """Stable radix sort of (int64-viewed-as-u64 keys, int32 values) on the device; returns new tensors."""
    _lib.require_cuda(keys, values)
    assert keys.dtype == torch.int64 and values.dtype == torch.int32 and keys.numel() == values.numel()
    dev = keys.device
    k, v = keys.clone().contiguous(), values.clone().contiguous()
    ka, va = torch.empty_like(k), torch.empty_like(v)
    n = k.numel()
    tb = int(_lib.lib().gvf_sort_tmp_bytes(n))
    tmp = torch.empty(tb + 256, dtype=torch.uint8, device=dev)
    base = (tmp.data_ptr() + 255) // 256 * 256
    rc = _lib.lib().gvf_sort_pairs_u64(_lib.ptr(k), _lib.ptr(ka), _lib.ptr(v), _lib.ptr(va), n, end_bit,
                                       ctypes.c_void_p(base), tb, _lib.current_stream(dev))
    _lib.check(rc, "gvf_sort_pairs_u64")
    return k, v


def frames_to_uint8(rgb: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """(.., H, W) float frames -> uint8 on the device: clamp(0,1) * 255 truncated, as the reference's
    render_and_save_images does on the host (utils/inference_utils.py:280-286)."""
    _lib.require_cuda(rgb)
    rgb = _f32c(rgb, "rgb")
    if out is None:
        out = torch.empty(rgb.shape, dtype=torch.uint8, device=rgb.device)
    _lib.check(_lib.lib().gvf_rgb_to_u8(_lib.ptr(rgb), _lib.ptr(out), rgb.numel(), _lib.current_stream(rgb.device)),
               "gvf_rgb_to_u8")
    return out
