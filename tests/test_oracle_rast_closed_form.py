"""Known answers of the published algorithm for the rasteriser oracle (oracle/rast_oracle.c), worked out by hand for scenes simple enough
to have a closed form -- 3D Gaussian Splatting (Kerbl et al. 2023, eq. 2-3 and appendix A/C: Sigma' = J W Sigma W^T J^T, alpha = o *
exp(-1/2 d^T Sigma'^-1 d), C = sum c_i alpha_i prod (1 - alpha_j), alpha clamped to 0.99, contributions under 1/255 skipped, stop at
T < 1e-4, pixel (x, y) sampled at ((ndc + 1) * W - 1) / 2) and the 2D Mip filter of Mip-Splatting (Yu et al. 2024, eq. 9: Sigma' + s I with
the opacity scaled by sqrt(|Sigma'| / |Sigma' + s I|)).  The reference's third-party CUDA extension is absent from the tree (parity
unpinned against its binaries: DESIGN 2.1); these cases pin the constants a restatement gets wrong first: the pixel-centre convention, the
two dilation modes, the compositing order, the three thresholds."""
import math

import numpy as np
import pytest
import torch

from gvfdiffusion_amd import synthetic
from rast_util import camera_block, oracle_render

H = W = 32
BG = (0.2, 0.4, 0.6)


def _scene(z_list, s_list, op_list, col_list, xy=(0.0, 0.0)):
    """Isotropic Gaussians on (or parallel to) the optical axis of the azi = 0, elev = 0 orbit camera, identity rotations, precomputed colours."""
    cam = camera_block(azi=0.0, elev=0.0, radius=2.0)
    Vw = cam["extrinsics"].double()                       # world -> camera
    pts_cam = torch.tensor([[xy[0], xy[1], z, 1.0] for z in z_list], dtype=torch.float64)
    pts_w = (torch.inverse(Vw) @ pts_cam.T).T[:, :3]
    n = len(z_list)
    attrs = dict(means3D=pts_w.float(), scales=torch.tensor([[s, s, s] for s in s_list], dtype=torch.float32),
                 rotations=torch.tensor([[1.0, 0, 0, 0]] * n), opacities=torch.tensor(op_list, dtype=torch.float32).reshape(n, 1),
                 shs=torch.zeros((n, 1, 3)))
    return cam, attrs, torch.tensor(col_list, dtype=torch.float32)


def _sigma2(cam, s, z, mode, kernel):
    f = W / (2 * cam["tanfovx"])
    base = (f * s / z) ** 2
    if mode == 0:
        return base + kernel, math.sqrt(base * base / ((base + kernel) ** 2))      # sqrt(|S| / |S + k I|) for S = base I
    return base + 0.3, 1.0


@pytest.mark.parametrize("mode", [0, 1])
def test_single_isotropic_gaussian_on_the_optical_axis(oracle_lib, mode):
    z, s, op, col = 2.0, 0.08, 0.7, (0.9, 0.3, 0.1)
    cam, attrs, c = _scene([z], [s], [op], [col])
    out = oracle_render(oracle_lib, attrs, cam, H, W, 0, mode=mode, kernel_size=0.1, bg=BG, colors_precomp=c)
    img, alpha = out["color"], out["alpha"]
    sig2, coef = _sigma2(cam, s, z, mode, 0.1)
    # the mean projects to ((0 + 1) * W - 1) / 2 = 15.5: BETWEEN pixels 15 and 16 -- the image is mirror symmetric about it
    assert np.allclose(img, img[:, :, ::-1], atol=1e-6) and np.allclose(img, img[:, ::-1, :], atol=1e-6)
    yy, xx = np.mgrid[0:H, 0:W]
    d2 = (xx - 15.5) ** 2 + (yy - 15.5) ** 2
    a = np.minimum(0.99, op * coef * np.exp(-0.5 * d2 / sig2))
    a[a < 1.0 / 255.0] = 0.0
    # (the oracle composites inside the 3-sigma rect's tiles only; with sig2 ~ 0.9 .. 1.1 px^2 alpha is < 1/255 long before the rect ends)
    assert np.abs(alpha - a).max() < 2e-6
    exp_img = np.asarray(col)[:, None, None] * a[None] + np.asarray(BG)[:, None, None] * (1 - a[None])
    assert np.abs(img - exp_img).max() < 2e-6
    assert abs(alpha[15, 15] - op * coef * math.exp(-0.25 / sig2)) < 1e-6           # the four centre pixels sit (0.5, 0.5) away
    assert np.allclose(img[:, 0, 0], BG, atol=0) and alpha[0, 0] == 0.0             # nothing splatted: exactly the background


def test_front_to_back_compositing_and_alpha_clamp(oracle_lib):
    """Two coincident footprints at different depths: C = c1 a1 + c2 a2 (1 - a1) + bg (1 - a1)(1 - a2), nearer first whatever the input order;
    opacity 1 clamps at 0.99."""
    z, s = [2.6, 1.4], [0.13, 0.07]                       # same s / z: identical footprints
    cam, attrs, c = _scene(z, s, [1.0, 0.6], [(0.0, 1.0, 0.0), (1.0, 0.0, 0.0)])
    out = oracle_render(oracle_lib, attrs, cam, H, W, 0, mode=1, bg=BG, colors_precomp=c)
    sig2, _ = _sigma2(cam, s[0], z[0], 1, 0.1)
    g = math.exp(-0.25 / sig2)
    a_far, a_near = min(0.99, 1.0 * g), 0.6 * g
    exp = np.array([1.0, 0, 0]) * a_near + np.array([0, 1.0, 0]) * a_far * (1 - a_near) + np.asarray(BG) * (1 - a_near) * (1 - a_far)
    assert np.abs(out["color"][:, 15, 15] - exp).max() < 2e-6
    assert abs(out["alpha"][15, 15] - (1 - (1 - a_near) * (1 - a_far))) < 2e-6
    # a huge, opaque splat: alpha is 0.99 wherever exp() is close enough to 1, never more
    cam, attrs, c = _scene([2.0], [0.8], [1.0], [(1.0, 1.0, 1.0)])
    out = oracle_render(oracle_lib, attrs, cam, H, W, 0, mode=1, bg=(0, 0, 0), colors_precomp=c)
    assert abs(out["alpha"].max() - 0.99) < 1e-6 and abs(out["alpha"][15, 15] - 0.99) < 1e-6


def test_thresholds_skip_small_contributions_and_stop_when_opaque(oracle_lib):
    # (1) alpha < 1/255 contributes nothing: opacity just under the threshold leaves the exact background, just over it does not
    for op, seen in ((0.9 / 255.0, False), (1.2 / 255.0, True)):
        cam, attrs, c = _scene([2.0], [0.5], [op], [(1.0, 0.0, 0.0)])
        out = oracle_render(oracle_lib, attrs, cam, H, W, 0, mode=1, bg=BG, colors_precomp=c)
        assert (out["alpha"].max() > 0) == seen
    # (2) the walk stops IN FRONT OF the splat that would push T under 1e-4 (that splat is not composited): layers 0.9, 0.99, 0.99, 0.99 ->
    # T = 0.1, 1e-3, then 1e-3 * 0.01 = 1e-5 < 1e-4 stops before the third: the blue and the white layer behind never show
    cam, attrs, c = _scene([1.2, 1.6, 2.0, 2.4], [0.6, 0.8, 1.0, 1.2], [0.9, 1.0, 1.0, 1.0],
                           [(1.0, 0, 0), (0, 1.0, 0), (0, 0, 1.0), (1.0, 1.0, 1.0)])
    out = oracle_render(oracle_lib, attrs, cam, H, W, 0, mode=1, bg=(0, 0, 0), colors_precomp=c)
    sig2, _ = _sigma2(cam, 0.6, 1.2, 1, 0.1)
    g = math.exp(-0.25 / sig2)
    assert g > 0.99
    a1 = 0.9 * g
    px = out["color"][:, 15, 15]
    assert abs(px[0] - a1) < 2e-6 and abs(px[1] - 0.99 * (1 - a1)) < 2e-6 and px[2] == 0.0
    assert abs(out["alpha"][15, 15] - (1 - (1 - a1) * 0.01)) < 2e-6


def test_off_axis_mean_lands_on_the_pixel_the_projection_formula_says(oracle_lib):
    """A small Gaussian whose mean projects exactly onto a pixel centre: that pixel gets opacity * coef (exp(0) = 1), its four neighbours equal
    values -- the sample point of pixel (x, y) is (x, y) in the ((ndc + 1) * W - 1) / 2 frame, not (x + 0.5, y + 0.5)."""
    cam0 = camera_block(azi=0.0, elev=0.0, radius=2.0)
    z = 2.0
    # ndc_x = x_cam / (z tan) ; want px = 20  ->  ndc = (2 * 20 + 1) / W - 1
    ndc = (2 * 20 + 1) / W - 1.0
    x_cam = ndc * z * cam0["tanfovx"]
    ndy = (2 * 9 + 1) / H - 1.0
    y_cam = ndy * z * cam0["tanfovy"]
    cam, attrs, c = _scene([z], [0.04], [0.8], [(0.5, 0.5, 0.5)], xy=(x_cam, y_cam))
    out = oracle_render(oracle_lib, attrs, cam, H, W, 0, mode=1, bg=(0, 0, 0), colors_precomp=c)
    a = out["alpha"]
    iy, ix = np.unravel_index(np.argmax(a), a.shape)
    assert (ix, iy) == (20, 9)
    # (the Jacobian of an off-axis point adds an anisotropic term of relative size (x / z)^2 ~ 3e-2 to the footprint: neighbours are compared
    # pairwise, the peak against exp(0))
    assert abs(a[9, 20] - 0.8) < 2e-4
    assert abs(a[9, 19] - a[9, 21]) < 2e-3 and abs(a[8, 20] - a[10, 20]) < 2e-3
