"""Host side of the rasteriser operator: tensor checks, workspace management, C-ABI calls.

Reference seam: renderers/gaussian_render.py:110-143 (settings) and :198-220 (operator call) --
the reference imports the two classes from the external CUDA packages `diff_gaussian_rasterization`
(mip-splatting fork) and `diff_gauss`; here they are backed by libgvf_hip.so (csrc/rast.hip).
Differentiable: when an input requires grad, rasterize() runs through _RasterizeFn (forward keeps its
workspace, backward calls gvf_rast_backward) -- upstream's _RasterizeGaussians autograd.Function.
"""
import ctypes
import os
from typing import Optional

import torch

from . import _lib

_WORKSPACES = {}  # (device index, stream handle) -> uint8 tensor, grown on demand


def _workspace(device, nbytes: int) -> torch.Tensor:
    """Scratch of the non-differentiable calls.  One buffer per (device, stream): a buffer is only ever touched by
    work queued on the stream it was allocated on, so two renders on different streams never share scratch, and
    replacing a buffer that is too small is ordered after its last use by the caching allocator (same stream)."""
    dev_index = device.index if device.index is not None else torch.cuda.current_device()
    key = (dev_index, int(torch.cuda.current_stream(device).cuda_stream))
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
              want_alpha_depth=False, means2D=None):
    """One frame, activated inputs (GaussianRasterizer.__call__).  Returns dict of device tensors.  Differentiable
    w.r.t. means3D, means2D, shs | colors_precomp, opacities, scales, rotations | cov3D_precomp when grad is enabled
    and one of them requires it."""
    diff = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in
                                           (means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp))
    if not diff:
        return _rasterize_impl(settings, frame, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp,
                               subpixel_offset, want_alpha_depth)
    _lib.require_cuda(means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, means2D)
    color, alpha, depth, radii = _RasterizeFn.apply(settings, frame, bool(want_alpha_depth), subpixel_offset, means3D, means2D,
                                                    shs, colors_precomp, opacities, scales, rotations, cov3D_precomp)
    return dict(color=color, alpha=alpha if want_alpha_depth else None, depth=depth if want_alpha_depth else None,
                radii=radii, num_rendered=None)


def _rasterize_impl(settings, frame, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3D_precomp=None, subpixel_offset=None, want_alpha_depth=False, private_workspace=False):
    """private_workspace: the call gets its own workspace tensor, returned in the dict (the backward pass reads the
    splat records / sorted lists / tile ranges the forward left there)."""
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
        ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev) if private_workspace else _workspace(dev, nbytes + 256)
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
    out = dict(color=color, alpha=alpha, depth=depth, radii=radii, num_rendered=n)
    if private_workspace:
        out.update(workspace=ws, workspace_bytes=nbytes, max_rendered=cap)
    return out


class _RasterizeFn(torch.autograd.Function):
    """autograd wrapper of one operator call (upstream: diff_gaussian_rasterization._RasterizeGaussians).
    Inputs that may carry gradients: means3D, means2D (receives the screen-space gradient, NDC units, z = 0),
    shs | colors_precomp, opacities, scales + rotations | cov3D_precomp.  Outputs: color, alpha, depth (the last
    two are zeros-like placeholders unless want_alpha_depth), radii."""

    @staticmethod
    def forward(ctx, settings, frame, want_ad, subpixel_offset, means3D, means2D, shs, colors_precomp, opacities, scales,
                rotations, cov3D_precomp):
        # what backward hands to the C ABI must be what forward rendered: dense fp32.  Normalise BEFORE saving (a
        # strided view such as x[:, :3] or a half-precision tensor would otherwise be read as dense fp32 in backward);
        # the gradients are cast back to each input's dtype on the way out.
        ctx.in_dtypes = tuple(None if t is None else t.dtype for t in
                              (means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp))
        means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp = (
            _f32c(t, "input") for t in (means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp))
        out = _rasterize_impl(settings, frame, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp,
                              subpixel_offset, want_ad, private_workspace=True)
        ctx.settings, ctx.frame, ctx.want_ad = settings, frame, want_ad
        ctx.ws, ctx.ws_bytes, ctx.cap = out["workspace"], out["workspace_bytes"], out["max_rendered"]
        ctx.has = (shs is not None, colors_precomp is not None, scales is not None, cov3D_precomp is not None,
                   means2D is not None)
        ctx.save_for_backward(*[t if t is not None else torch.empty(0) for t in
                                (means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, subpixel_offset)])
        ctx.shapes = (opacities.shape, None if means2D is None else means2D.shape)
        dev = means3D.device
        H, W = settings.image_height, settings.image_width
        alpha = out["alpha"] if want_ad else torch.zeros((H, W), device=dev)
        depth = out["depth"] if want_ad else torch.zeros((H, W), device=dev)
        ctx.mark_non_differentiable(out["radii"])
        ctx.num_rendered = out["num_rendered"]
        return out["color"], alpha, depth, out["radii"]

    @staticmethod
    def backward(ctx, g_color, g_alpha, g_depth, _g_radii):
        means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, sub = \
            [t if t.numel() > 0 or k == 0 else None for k, t in enumerate(ctx.saved_tensors)]
        has_shs, has_col, has_sr, has_cov, has_m2 = ctx.has
        shs = shs if has_shs else None
        colors_precomp = colors_precomp if has_col else None
        scales, rotations = (scales, rotations) if has_sr else (None, None)
        cov3D_precomp = cov3D_precomp if has_cov else None
        sub = sub if (sub is not None and sub.numel() > 0) else None
        dev = means3D.device
        P = means3D.shape[0]
        M = 0 if shs is None else shs.shape[1]
        f32 = dict(dtype=torch.float32, device=dev)
        g_color = _f32c(g_color if g_color is not None else torch.zeros((3, ctx.settings.image_height, ctx.settings.image_width), **f32), "g")
        g_alpha = _f32c(g_alpha, "g") if ctx.want_ad and g_alpha is not None else None
        g_depth = _f32c(g_depth, "g") if ctx.want_ad and g_depth is not None else None
        nb = ctypes.c_size_t(0)
        _lib.check(_lib.lib().gvf_rast_backward_scratch_bytes(P, ctypes.byref(nb)), "gvf_rast_backward_scratch_bytes")
        scratch = torch.empty(int(nb.value) + 16, dtype=torch.uint8, device=dev)
        sbase = (scratch.data_ptr() + 15) // 16 * 16
        d_m3 = torch.empty((P, 3), **f32)
        d_m2 = torch.empty((P, 2), **f32)
        d_shs = torch.empty((P, M, 3), **f32) if shs is not None else None
        d_col = torch.empty((P, 3), **f32) if colors_precomp is not None else None
        d_op = torch.empty((P,), **f32)
        d_sc = torch.empty((P, 3), **f32) if scales is not None else None
        d_ro = torch.empty((P, 4), **f32) if rotations is not None else None
        d_c6 = torch.empty((P, 6), **f32) if cov3D_precomp is not None else None
        base = (ctx.ws.data_ptr() + 255) // 256 * 256
        rc = _lib.lib().gvf_rast_backward(
            ctypes.byref(ctx.settings), ctypes.byref(ctx.frame), P, M, _lib.ptr(means3D), _lib.ptr(shs), _lib.ptr(colors_precomp),
            _lib.ptr(opacities), _lib.ptr(scales), _lib.ptr(rotations), _lib.ptr(cov3D_precomp), _lib.ptr(sub),
            ctypes.c_void_p(base), ctx.ws_bytes, ctx.cap, _lib.ptr(g_color), _lib.ptr(g_alpha), _lib.ptr(g_depth),
            ctypes.c_void_p(sbase), int(nb.value), _lib.ptr(d_m3), _lib.ptr(d_m2), _lib.ptr(d_shs), _lib.ptr(d_col),
            _lib.ptr(d_op), _lib.ptr(d_sc), _lib.ptr(d_ro), _lib.ptr(d_c6), _lib.current_stream(dev))
        _lib.check(rc, "gvf_rast_backward")
        op_shape, m2_shape = ctx.shapes
        g_m2 = None
        if has_m2:
            g_m2 = torch.zeros(m2_shape, **f32)
            g_m2[:, :2] = d_m2
        grads = (d_m3, g_m2, d_shs, d_col, d_op.reshape(op_shape), d_sc, d_ro, d_c6)
        grads = tuple(g if g is None or dt is None or g.dtype == dt else g.to(dt) for g, dt in zip(grads, ctx.in_dtypes))
        return (None, None, None, None) + grads


def rasterize_batched(settings, frames, act, xyz_raw, features_dc, scaling_raw, rotation_raw, opacity_raw,
                      delta=None, want_alpha_depth=False, want_radii=False, max_rendered=None, sync=True, color_u8=False):
    """F frames in one call with GaussianModel activations + per-frame deltas fused in-kernel.

    color_u8: `color` comes back as uint8 (F, 3, H, W) = clamp(rgb, 0, 1) * 255 truncated -- the reference's frame post-process
    (utils/inference_utils.py:280-286) in the compositing kernel's epilogue (gvf_rast_forward_batched_u8; bit-identical to frames_to_uint8 of the
    fp32 frames; no alpha / depth / radii then).

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
    if color_u8 and (want_alpha_depth or want_radii):
        raise _lib.GvfError("color_u8: the uint8 entry point has no alpha / depth / radii outputs")
    color = torch.empty((F, 3, H, W), dtype=torch.uint8 if color_u8 else torch.float32, device=dev)
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
        if color_u8:
            rc = _lib.lib().gvf_rast_forward_batched_u8(
                ctypes.byref(settings), arr, F, ctypes.byref(act), P, M, _lib.ptr(xyz_raw), _lib.ptr(features_dc),
                _lib.ptr(scaling_raw), _lib.ptr(rotation_raw), _lib.ptr(opacity_raw), _lib.ptr(delta), n_delta,
                ctypes.c_void_p(base), nbytes, cap, _lib.ptr(color), _lib.ptr(nr), _lib.current_stream(dev))
        else:
            rc = _lib.lib().gvf_rast_forward_batched(
                ctypes.byref(settings), arr, F, ctypes.byref(act), P, M, _lib.ptr(xyz_raw), _lib.ptr(features_dc),
                _lib.ptr(scaling_raw), _lib.ptr(rotation_raw), _lib.ptr(opacity_raw), _lib.ptr(delta), n_delta,
                ctypes.c_void_p(base), nbytes, cap, _lib.ptr(color), _lib.ptr(alpha), _lib.ptr(depth),
                _lib.ptr(radii), _lib.ptr(nr), _lib.current_stream(dev))
        _lib.check(rc, "gvf_rast_forward_batched")
        _LAST_CARVE[(dev.index if dev.index is not None else torch.cuda.current_device(), int(torch.cuda.current_stream(dev).cuda_stream))] = \
            (ws, base, nbytes, P, F, H, W, cap)         # `ws`: the tensor itself, so that the address stays this workspace's
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


_LAST_CARVE = {}     # (device index, stream handle) -> the workspace TENSOR + arguments of the last batched call there (sort_class_counts).
#                      Holding the tensor pins its storage: a bare address could belong to another tensor by the time the diagnostic reads it.


def sort_class_counts(device=None):
    """(segments with 1537 .. 16384 keys, segments with more) of the LAST rasterize_batched() call on the current stream: how many
    (frame, tile) segments went through the per-tile sort's LDS launches / its in-place HBM class (gvf_rast_sort_class_counts; a test
    diagnostic -- it waits for the stream)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), int(torch.cuda.current_stream(device).cuda_stream))
    if key not in _LAST_CARVE:
        raise _lib.GvfError("no batched rasteriser call on this stream yet")
    _ws, base, nbytes, P, F, H, W, cap = _LAST_CARVE[key]
    out = (ctypes.c_uint32 * 2)()
    _lib.check(_lib.lib().gvf_rast_sort_class_counts(ctypes.c_void_p(base), nbytes, P, F, H, W, cap, out, _lib.current_stream(device)),
               "gvf_rast_sort_class_counts")
    return int(out[0]), int(out[1])


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


def tile_sort_u64(keys: torch.Tensor, ranges: torch.Tensor) -> torch.Tensor:
    """Per-segment sort of (depth bits << 32 | id) keys, the per-tile half of the rasteriser's sort stage (gvf_tile_sort_u64):
    ranges is (nseg, 2) int32 [begin, end); returns the ids (low key words) in sorted order at the same positions."""
    _lib.require_cuda(keys, ranges)
    assert keys.dtype == torch.int64 and ranges.dtype == torch.int32 and ranges.dim() == 2 and ranges.shape[1] == 2
    k, r = keys.clone().contiguous(), ranges.contiguous()
    ids = torch.full((k.numel(),), -1, dtype=torch.int32, device=k.device)
    scratch = torch.empty((2 + 2 * r.shape[0],), dtype=torch.int32, device=k.device)
    _lib.check(_lib.lib().gvf_tile_sort_u64(_lib.ptr(k), _lib.ptr(r), r.shape[0], _lib.ptr(ids), _lib.ptr(scratch),
                                            _lib.current_stream(k.device)), "gvf_tile_sort_u64")
    return ids


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
