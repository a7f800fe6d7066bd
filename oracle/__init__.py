"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package
(never gvfdiffusion_amd/).  It wraps the plain-C restatements in this directory
(rast_oracle.c, rast_bwd_oracle.c, vox2seq_oracle.c -> libgvf_oracle.so, built by `make -C oracle`) and holds the
torch-fp32 restatements of the floating-point DiT / sampler path (dit_ref.py, dpm_ref.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libgvf_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("rast_oracle.c", "rast_bwd_oracle.c", "vox2seq_oracle.c", "Makefile")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgvf_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, ty=ctypes.c_float):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ty))


def num_threads() -> int:
    return int(lib().gvfo_num_threads())


def _common(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp):
    means3D = _f32(means3D)
    P = means3D.shape[0]
    shs = _f32(shs)
    M = 0 if shs is None else shs.shape[1]
    return (P, M, means3D, shs, _f32(colors_precomp), _f32(np.reshape(opacities, (-1,))), _f32(scales),
            _f32(rotations), _f32(cov3D_precomp))


def rast_preprocess(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, *, H, W,
                    tanfovx, tanfovy, kernel_size, scale_modifier, mode, viewmatrix, projmatrix, campos,
                    sh_degree, tight=False):
    lib().gvfo_set_tight_binning(int(bool(tight)))
    P, M, m3, sh, cp, op, sc, ro, c3 = _common(means3D, shs, colors_precomp, opacities, scales, rotations,
                                               cov3D_precomp)
    geom = np.zeros((P, 24), np.float32)
    v, pj, cam = _f32(np.reshape(viewmatrix, (-1,))), _f32(np.reshape(projmatrix, (-1,))), _f32(campos)
    rc = lib().gvfo_preprocess(P, M, int(sh_degree), _ptr(m3), _ptr(sh), _ptr(cp), _ptr(op), _ptr(sc), _ptr(ro),
                               _ptr(c3), int(H), int(W), ctypes.c_float(tanfovx), ctypes.c_float(tanfovy),
                               ctypes.c_float(kernel_size), ctypes.c_float(scale_modifier), int(mode), _ptr(v),
                               _ptr(pj), _ptr(cam), _ptr(geom))
    lib().gvfo_set_tight_binning(0)
    assert rc == 0
    return geom


def rast_render(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, *, H, W, tanfovx,
                tanfovy, kernel_size, scale_modifier, mode, viewmatrix, projmatrix, campos, sh_degree, bg,
                subpixel_offset=None, nthreads=0, brute=False, tight=False):
    """Returns dict(color[3,H,W], alpha[H,W], depth[H,W], radii[P], num_rendered, flags[H,W]).
    tight=True: drop (Gaussian, tile) instances that cannot reach alpha 1/255 anywhere in the tile (the HIP
    path's default binning; same image, smaller num_rendered) instead of upstream's 3-sigma rect."""
    lib().gvfo_set_tight_binning(int(bool(tight)))
    P, M, m3, sh, cp, op, sc, ro, c3 = _common(means3D, shs, colors_precomp, opacities, scales, rotations,
                                               cov3D_precomp)
    v, pj, cam = _f32(np.reshape(viewmatrix, (-1,))), _f32(np.reshape(projmatrix, (-1,))), _f32(campos)
    bg = _f32(bg)
    color = np.zeros((3, H, W), np.float32)
    alpha = np.zeros((H, W), np.float32)
    depth = np.zeros((H, W), np.float32)
    radii = np.zeros((P,), np.int32)
    nr = np.zeros((1,), np.uint32)
    flags = np.zeros((H, W), np.uint8)
    f = ctypes.c_float
    if brute:
        rc = lib().gvfo_render_brute(P, M, int(sh_degree), _ptr(m3), _ptr(sh), _ptr(cp), _ptr(op), _ptr(sc),
                                     _ptr(ro), _ptr(c3), int(H), int(W), f(tanfovx), f(tanfovy), f(kernel_size),
                                     f(scale_modifier), int(mode), _ptr(v), _ptr(pj), _ptr(cam), _ptr(bg),
                                     _ptr(color), _ptr(alpha), _ptr(depth))
    else:
        so = _f32(subpixel_offset)
        rc = lib().gvfo_render(P, M, int(sh_degree), _ptr(m3), _ptr(sh), _ptr(cp), _ptr(op), _ptr(sc), _ptr(ro),
                               _ptr(c3), _ptr(so), int(H), int(W), f(tanfovx), f(tanfovy), f(kernel_size),
                               f(scale_modifier), int(mode), _ptr(v), _ptr(pj), _ptr(cam), _ptr(bg), _ptr(color),
                               _ptr(alpha), _ptr(depth), _ptr(radii, ctypes.c_int32), _ptr(nr, ctypes.c_uint32),
                               _ptr(flags, ctypes.c_uint8), int(nthreads))
    lib().gvfo_set_tight_binning(0)
    assert rc == 0, rc
    return dict(color=color, alpha=alpha, depth=depth, radii=radii, num_rendered=int(nr[0]), flags=flags)


def gaussian_activate(xyz_raw, features_dc, scaling_raw, rotation_raw, opacity_raw, delta, *, aabb, scale_bias,
                      opacity_bias, min_kernel_size, scaling_activation):
    xyz_raw = _f32(xyz_raw); P = xyz_raw.shape[0]
    features_dc = _f32(features_dc); M = features_dc.shape[1]
    scaling_raw, rotation_raw = _f32(scaling_raw), _f32(rotation_raw)
    opacity_raw = _f32(np.reshape(opacity_raw, (-1,)))
    delta = _f32(delta)
    aabb = _f32(aabb)
    out = dict(means3D=np.zeros((P, 3), np.float32), scales=np.zeros((P, 3), np.float32),
               rotations=np.zeros((P, 4), np.float32), shs=np.zeros((P, M, 3), np.float32),
               opacities=np.zeros((P,), np.float32))
    f = ctypes.c_float
    rc = lib().gvfo_activate(P, M, _ptr(aabb), f(scale_bias), f(opacity_bias), f(min_kernel_size),
                             int(scaling_activation), _ptr(xyz_raw), _ptr(features_dc), _ptr(scaling_raw),
                             _ptr(rotation_raw), _ptr(opacity_raw), _ptr(delta), _ptr(out["means3D"]),
                             _ptr(out["scales"]), _ptr(out["rotations"]), _ptr(out["shs"]), _ptr(out["opacities"]))
    assert rc == 0
    return out


def _vox(fn, a, b, c, n_out):
    a = np.ascontiguousarray(a, np.int32); n = a.shape[0]
    ip = lambda x: x.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    if n_out == 1:
        b = np.ascontiguousarray(b, np.int32); c = np.ascontiguousarray(c, np.int32)
        out = np.zeros((n,), np.int32)
        fn(ctypes.c_int64(n), ip(a), ip(b), ip(c), ip(out))
        return out
    x, y, z = (np.zeros((n,), np.int32) for _ in range(3))
    fn(ctypes.c_int64(n), ip(a), ip(x), ip(y), ip(z))
    return np.stack([x, y, z], -1)


def vox2seq_encode(coords, mode="z_order"):
    """coords (N,3) int -> codes (N,) int32.  mode: 'z_order' | 'hilbert'."""
    coords = np.asarray(coords)
    fn = lib().gvfo_z_order_encode if mode == "z_order" else lib().gvfo_hilbert_encode
    return _vox(fn, coords[:, 0], coords[:, 1], coords[:, 2], 1)


def vox2seq_decode(codes, mode="z_order"):
    fn = lib().gvfo_z_order_decode if mode == "z_order" else lib().gvfo_hilbert_decode
    return _vox(fn, codes, None, None, 3)


# ---- double-precision forward + backward of the rasteriser operator (rast_bwd_oracle.c) -----------------------
def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _p64(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _scene64(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, H, W, tanfovx, tanfovy,
             kernel_size, scale_modifier, mode, viewmatrix, projmatrix, campos, sh_degree, bg):
    m3 = _f64(means3D); P = m3.shape[0]
    sh = _f64(shs); M = 0 if sh is None else sh.shape[1]
    keep = (m3, sh, _f64(colors_precomp), _f64(np.reshape(opacities, (-1,))), _f64(scales), _f64(rotations),
            _f64(cov3D_precomp), _f64(np.reshape(viewmatrix, (-1,))), _f64(np.reshape(projmatrix, (-1,))), _f64(campos), _f64(bg))
    d = ctypes.c_double
    args = [P, M, int(sh_degree)] + [_p64(a) for a in keep[:7]] + [int(H), int(W), d(tanfovx), d(tanfovy), d(kernel_size),
                                                                   d(scale_modifier), int(mode)] + [_p64(a) for a in keep[7:]]
    return P, M, keep, args


def rast64_forward(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, *, H, W, tanfovx, tanfovy,
                   kernel_size, scale_modifier, mode, viewmatrix, projmatrix, campos, sh_degree, bg):
    """Double-precision forward: dict(color[3,H,W], alpha[H,W], depth[H,W])."""
    P, M, keep, args = _scene64(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, H, W, tanfovx,
                                tanfovy, kernel_size, scale_modifier, mode, viewmatrix, projmatrix, campos, sh_degree, bg)
    color, alpha, depth = np.zeros((3, H, W)), np.zeros((H, W)), np.zeros((H, W))
    rc = lib().gvfo64_forward(*args, _p64(color), _p64(alpha), _p64(depth))
    assert rc == 0
    return dict(color=color, alpha=alpha, depth=depth)


def rast64_backward(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, dL_dcolor, dL_dalpha=None,
                    dL_ddepth=None, *, H, W, tanfovx, tanfovy, kernel_size, scale_modifier, mode, viewmatrix, projmatrix,
                    campos, sh_degree, bg):
    """Double-precision gradients of sum(dL_dcolor*color) + sum(dL_dalpha*alpha) + sum(dL_ddepth*depth):
    dict(means3D, means2D [NDC units], shs | colors_precomp, opacities, scales, rotations | cov3D_precomp)."""
    P, M, keep, args = _scene64(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, H, W, tanfovx,
                                tanfovy, kernel_size, scale_modifier, mode, viewmatrix, projmatrix, campos, sh_degree, bg)
    gc, ga, gd = _f64(dL_dcolor), _f64(dL_dalpha), _f64(dL_ddepth)
    out = dict(means3D=np.zeros((P, 3)), means2D=np.zeros((P, 2)), opacities=np.zeros((P,)))
    g_shs = np.zeros((P, M, 3)) if shs is not None else None
    g_col = np.zeros((P, 3)) if colors_precomp is not None else None
    g_sc = np.zeros((P, 3)) if cov3D_precomp is None else None
    g_ro = np.zeros((P, 4)) if cov3D_precomp is None else None
    g_c6 = np.zeros((P, 6))
    rc = lib().gvfo64_backward(*args, _p64(gc), _p64(ga), _p64(gd), _p64(out["means3D"]), _p64(out["means2D"]), _p64(g_shs),
                               _p64(g_col), _p64(out["opacities"]), _p64(g_sc), _p64(g_ro), _p64(g_c6))
    assert rc == 0
    out.update(shs=g_shs, colors_precomp=g_col, scales=g_sc, rotations=g_ro, cov3D_precomp=g_c6)
    return out
