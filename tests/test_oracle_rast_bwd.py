"""CPU tests of the double-precision rasteriser oracle with backward pass (oracle/rast_bwd_oracle.c):
its forward against the float forward oracle, and every gradient it returns against central finite
differences of its own forward (the mathematics that pins it -- the reference holds no gradient test for the
third-party operator)."""
import numpy as np
import pytest
import torch

import oracle
from gvfdiffusion_amd import synthetic
from rast_util import camera_block, oracle_render


def scene(P, deg, seed, S):
    a = synthetic.random_gaussians(P, sh_degree=deg, seed=seed, scale_lo=0.01, scale_hi=0.06)
    a["means3D"] = a["means3D"] * 0.7                       # keep every Gaussian inside the 1.3 tan(fov) clamp
    a["opacities"] = a["opacities"].clamp(0.05, 0.9)         # alpha never reaches the 0.99 clamp
    cam = camera_block(azi=25.0, elev=10.0)
    kw = dict(H=S, W=S, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], kernel_size=synthetic.KERNEL_2D, scale_modifier=1.0,
              viewmatrix=cam["viewmatrix"].numpy(), projmatrix=cam["projmatrix"].numpy(), campos=cam["campos"].numpy(),
              sh_degree=deg, bg=np.asarray([0.2, 0.5, 0.9]))
    return a, cam, kw


@pytest.mark.parametrize("mode", [0, 1])
def test_double_forward_matches_float_oracle(mode):
    S, deg = 48, 2
    a, cam, kw = scene(300, deg, 3, S)
    n = lambda t: t.numpy()
    out = oracle.rast64_forward(n(a["means3D"]), n(a["shs"]), None, n(a["opacities"]), n(a["scales"]), n(a["rotations"]), None,
                                mode=mode, **kw)
    ref = oracle_render(oracle, a, cam, S, S, deg, mode=mode, bg=(0.2, 0.5, 0.9))
    clean = ref["flags"] == 0
    assert clean.mean() > 0.97
    assert np.abs(out["color"] - ref["color"]).max(axis=0)[clean].max() < 2e-5
    assert np.abs(out["alpha"] - ref["alpha"])[clean].max() < 2e-5
    assert np.abs(out["depth"] - ref["depth"])[clean].max() < 5e-5


def _loss_weights(S, seed):
    g = np.random.default_rng(seed)
    return g.standard_normal((3, S, S)), g.standard_normal((S, S)), g.standard_normal((S, S))


def _loss(out, wc, wa, wd, mode):
    v = float((out["color"] * wc).sum())
    if mode == 1:
        v += float((out["alpha"] * wa).sum() + (out["depth"] * wd).sum())
    return v


@pytest.mark.parametrize("mode,deg,use_cov,use_rgb", [(0, 2, False, False), (1, 1, False, False), (0, 3, False, False),
                                                      (0, 0, True, True), (1, 0, False, True)])
def test_backward_matches_finite_differences(mode, deg, use_cov, use_rgb):
    S, P = 32, 40
    a, cam, kw = scene(P, deg, 11 + mode + deg, S)
    n = lambda t: t.double().numpy().copy()
    inputs = dict(means3D=n(a["means3D"]), shs=None if use_rgb else n(a["shs"]),
                  colors_precomp=np.random.default_rng(5).random((P, 3)) if use_rgb else None,
                  opacities=n(a["opacities"]).reshape(-1), scales=None, rotations=None, cov3D_precomp=None)
    if use_cov:
        r, x, y, z = n(a["rotations"]).T
        R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                      2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
        L = R * n(a["scales"])[:, None, :]
        Sg = L @ L.transpose(0, 2, 1)
        inputs["cov3D_precomp"] = np.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]], 1)
    else:
        inputs["scales"], inputs["rotations"] = n(a["scales"]), n(a["rotations"])
    order = ("means3D", "shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp")
    wc, wa, wd = _loss_weights(S, 0)

    def f():
        return _loss(oracle.rast64_forward(*[inputs[k] for k in order], mode=mode, **kw), wc, wa, wd, mode)

    grads = oracle.rast64_backward(*[inputs[k] for k in order], wc, wa if mode == 1 else None, wd if mode == 1 else None,
                                   mode=mode, **kw)
    rng = np.random.default_rng(1)
    checked = jumps = 0
    for name in order:
        x = inputs[name]
        if x is None:
            continue
        g = grads[name].reshape(x.shape)
        flat = x.reshape(-1)
        idx = rng.choice(flat.size, size=min(flat.size, 40), replace=False)
        scale = max(1e-12, float(np.abs(g).max()))
        for i in idx:
            # the off-diagonal entries of a symmetric cov3D are stored once: their derivative counts both copies
            eps = 1e-6 * max(1.0, abs(flat[i])) if name != "cov3D_precomp" else 1e-9
            old = flat[i]
            fds = []
            for e in (eps, eps / 8):
                flat[i] = old + e; fp = f()
                flat[i] = old - e; fm = f()
                fds.append((fp - fm) / (2 * e))
            flat[i] = old
            got = g.reshape(-1)[i]
            if abs(fds[0] - fds[1]) > 1e-3 * scale + 1e-6:
                # the forward is piecewise smooth (radius ceil -> tile rect, alpha < 1/255, T < 1e-4): this interval
                # straddles a jump, where a finite difference says nothing about the derivative
                jumps += 1
                continue
            assert abs(fds[1] - got) <= 2e-4 * scale + 1e-7, f"{name}[{i}]: finite difference {fds[1]} vs backward {got} (scale {scale})"
            checked += 1
    assert checked > 100 and jumps <= 0.05 * checked
    # the screen-space gradient is the pixel-space one in NDC units: cross-check through means3D on a pure translation
    assert np.isfinite(grads["means2D"]).all() and np.abs(grads["means2D"]).max() > 0


@pytest.mark.parametrize("mode", [0, 1])
def test_backward_closed_form_single_gaussian(mode):
    """One isotropic Gaussian on the optical axis (the scene of tests/test_oracle_rast_closed_form.py): colour = c a + bg (1 - a) with
    a = o * coef * g(x), so for the loss sum(w * colour)   dL/dc_k = sum_x w_k a   and   dL/do = sum_x sum_k w_k (c_k - bg_k) coef g
    over the pixels the splat reaches (a >= 1/255, no clamp at o = 0.7) -- worked out by hand, no finite differences; dL/d(scale) by
    symmetry is the same for the two image-plane axes and dL/d(mean_x) = dL/d(mean_y) = 0 for a symmetric weight."""
    import math
    import test_oracle_rast_closed_form as cf
    H, W = cf.H, cf.W
    z, s, op, col, bg = 2.0, 0.08, 0.7, np.array([0.9, 0.3, 0.1]), np.array([0.2, 0.4, 0.6])
    cam, attrs, c = cf._scene([z], [s], [op], [tuple(col)])
    kw = dict(H=H, W=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], kernel_size=0.1, scale_modifier=1.0,
              viewmatrix=cam["viewmatrix"].numpy(), projmatrix=cam["projmatrix"].numpy(), campos=cam["campos"].numpy(), sh_degree=0, bg=bg)
    n = lambda t: t.double().numpy().copy()
    yy, xx = np.mgrid[0:H, 0:W]
    d2 = (xx - 15.5) ** 2 + (yy - 15.5) ** 2
    wgt = np.stack([1.0 + 0.01 * d2, 2.0 - 0.02 * d2, 0.5 + 0.0 * d2])           # radially symmetric weights, different per channel
    g = oracle.rast64_backward(n(attrs["means3D"]), None, n(c), n(attrs["opacities"]).reshape(-1), n(attrs["scales"]), n(attrs["rotations"]), None,
                               wgt, mode=mode, **kw)
    sig2, coef = cf._sigma2(cam, s, z, mode, 0.1)
    gx = np.exp(-0.5 * d2 / sig2)
    a = op * coef * gx
    reach = a >= 1.0 / 255.0
    exp_dcol = (wgt * (a * reach)[None]).sum(axis=(1, 2))
    exp_dop = float(((wgt * (col - bg)[:, None, None]).sum(axis=0) * coef * gx * reach).sum())
    assert np.allclose(g["colors_precomp"][0], exp_dcol, rtol=1e-6, atol=1e-9)
    assert abs(g["opacities"][0] - exp_dop) < 1e-6 * max(1.0, abs(exp_dop))
    assert np.abs(g["means2D"][0]).max() < 1e-9 * max(1.0, abs(exp_dop))          # symmetric scene, symmetric weights: no pull on the mean
    # the camera looks along one world axis: the two scale components that span the image plane get equal gradients, and the sign says
    # "a bigger splat covers more of the brighter-than-background red / darker blue": checked against the closed-form derivative in sigma^2
    base = sig2 - (0.1 if mode == 0 else 0.3)
    dcoef = 0.0 if mode == 1 else (0.1 / (base + 0.1) ** 2)                        # d coef / d base, coef = base / (base + k)
    da_dbase = op * (dcoef * gx + coef * gx * (0.5 * d2 / sig2 ** 2))
    dL_dbase = float(((wgt * (col - bg)[:, None, None]).sum(axis=0) * da_dbase * reach).sum())
    # base = (f s / z)^2 is the variance along BOTH image axes; an isotropic change of the scale moves both: dL/ds_u + dL/ds_v = dL/dbase * 2 base / s
    # ... with Sigma' = diag(base_u, base_v): a = o coef(base_u, base_v) exp(-dx^2 / 2 su2 - dy^2 / 2 sv2); by symmetry each axis carries half
    gs = np.sort(np.abs(g["scales"][0]))
    assert abs(gs[2] - gs[1]) < 1e-6 * max(1.0, gs[2])                             # the two image-plane axes agree
    assert abs((gs[1] + gs[2]) - abs(dL_dbase * 2 * base / s)) < 2e-5 * max(1.0, abs(dL_dbase * 2 * base / s))
