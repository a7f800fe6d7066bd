"""Shared by tests/golden/make_golden.py::gen_align (which drives the REFERENCE's align_gaussian_to_canonical) and
tests/test_align_golden.py (which drives gvfdiffusion_amd's): a deterministic stand-in for the renderer -- the image an
"object" shows from azimuth index v -- so that everything between the renderer's output and the function's return value
(bounding boxes, scale factor, bicubic resize, pad / crop, L1 score, arg-min, rotation of positions and quaternions) is
pinned to the reference without a GPU."""
import math

import torch

SIZE = 512          # hard-coded in the reference (utils/inference_utils.py:96-110)


def view(v: int, n_views: int):
    """-> rgb (3,512,512) in [0,1], alpha (512,512): a tilted ellipse whose extent and colours depend on the azimuth index."""
    a = 2 * math.pi * v / n_views
    yy, xx = torch.meshgrid(torch.arange(SIZE, dtype=torch.float32), torch.arange(SIZE, dtype=torch.float32), indexing="ij")
    cx, cy = 256 + 18 * math.sin(a), 256 + 9 * math.cos(2 * a)
    rx, ry = 95 + 40 * math.cos(a) ** 2, 70 + 25 * math.sin(a + 0.7) ** 2
    th = 0.5 * math.sin(a)
    dx, dy = xx - cx, yy - cy
    u = (dx * math.cos(th) + dy * math.sin(th)) / rx
    w = (-dx * math.sin(th) + dy * math.cos(th)) / ry
    r2 = u * u + w * w
    alpha = torch.clamp(1.6 - 1.6 * r2, 0.0, 1.0)
    base = torch.stack([0.5 + 0.5 * torch.sin(0.031 * xx + a), 0.5 + 0.5 * torch.cos(0.027 * yy - 2 * a),
                        0.5 + 0.5 * torch.sin(0.019 * (xx + yy) + 3 * a)])
    rgb = base * alpha + (1.0 - alpha)                 # white background, as the renderer composites
    return rgb, alpha


def canonical(v_star: int, n_views: int, zoom: float):
    """The 'photo': view v_star magnified by `zoom` about the image centre (bilinear), with its alpha."""
    rgb, alpha = view(v_star, n_views)
    t = int(round(SIZE * zoom))
    img = torch.nn.functional.interpolate(torch.cat([rgb, alpha[None]])[None], size=(t, t), mode="bilinear", align_corners=False)[0]
    if t >= SIZE:
        o = (t - SIZE) // 2
        img = img[:, o:o + SIZE, o:o + SIZE]
    else:                                              # zoomed out: white (alpha 0) margin
        o = (SIZE - t) // 2
        full = torch.cat([torch.ones((3, SIZE, SIZE)), torch.zeros((1, SIZE, SIZE))])
        full[:, o:o + t, o:o + t] = img
        img = full
    return img[:3].clamp(0, 1).contiguous(), img[3].contiguous()


class ToyGaussians:
    """The four accessors align_gaussian_to_canonical touches."""

    def __init__(self, n=40, seed=0):
        g = torch.Generator().manual_seed(seed)
        self._xyz = torch.rand((n, 3), generator=g) - 0.5
        self._rot = torch.nn.functional.normalize(torch.randn((n, 4), generator=g), dim=1)

    get_xyz = property(lambda self: self._xyz)
    get_rotation = property(lambda self: self._rot)

    def from_xyz(self, x):
        self._xyz = x

    def from_rotation(self, r):
        self._rot = r
