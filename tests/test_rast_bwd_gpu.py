"""Backward pass of the HIP rasteriser operator (gvf_rast_backward through torch.autograd) against the double-precision
oracle (oracle/rast_bwd_oracle.c, itself pinned by finite differences in tests/test_oracle_rast_bwd.py) -- needs an MI355X."""
import numpy as np
import pytest
import torch

import oracle
from gvfdiffusion_amd import synthetic
from rast_util import camera_block
from test_rast_gpu import _settings

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(1e-30, np.linalg.norm(b)))


def _scene(P, deg, seed):
    a = synthetic.random_gaussians(P, sh_degree=deg, seed=seed, scale_lo=0.004, scale_hi=0.04)
    a["means3D"] = a["means3D"] * 0.8
    a["opacities"] = a["opacities"].clamp(0.02, 0.95)
    return a


@pytest.mark.parametrize("mode,deg,H,W,P", [(0, 2, 128, 128, 3000), (1, 1, 96, 144, 2000), (0, 3, 64, 64, 500), (0, 0, 200, 200, 8000)])
def test_gradients_match_double_oracle(cuda, mode, deg, H, W, P):
    a = _scene(P, deg, 5 + deg + mode)
    cam = camera_block(azi=30.0 + 20 * deg, elev=8.0)
    g = torch.Generator().manual_seed(3)
    wc = torch.randn((3, H, W), generator=g)
    wa, wd = torch.randn((H, W), generator=g), torch.randn((H, W), generator=g)
    leaves = {k: v.clone().to(cuda).requires_grad_(True) for k, v in a.items()}
    m2 = torch.zeros((P, 3), device=cuda, requires_grad=True)
    rast = _settings(cam, H, W, deg, mode, cuda, bg=(0.1, 0.4, 0.8))
    ret = rast(means3D=leaves["means3D"], means2D=m2, shs=leaves["shs"], colors_precomp=None, opacities=leaves["opacities"],
               scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None)
    loss = (ret[0] * wc.to(cuda)).sum()
    if mode == 1:
        loss = loss + (ret[3][0] * wa.to(cuda)).sum() + (ret[1][0] * wd.to(cuda)).sum()
    loss.backward()
    n = lambda t: t.detach().double().cpu().numpy()
    kw = dict(H=H, W=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], kernel_size=synthetic.KERNEL_2D, scale_modifier=1.0,
              viewmatrix=cam["viewmatrix"].numpy(), projmatrix=cam["projmatrix"].numpy(), campos=cam["campos"].numpy(),
              sh_degree=deg, bg=np.asarray([0.1, 0.4, 0.8]), mode=mode)
    ref = oracle.rast64_backward(n(a["means3D"]), n(a["shs"]), None, n(a["opacities"]), n(a["scales"]), n(a["rotations"]), None,
                                 n(wc), n(wa) if mode == 1 else None, n(wd) if mode == 1 else None, **kw)
    # float32 accumulation (atomics in arbitrary order) against float64: relative L2 per tensor
    errs = {k: rel(n(leaves[k].grad).reshape(ref[k].shape), ref[k]) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    errs["means2D"] = rel(n(m2.grad)[:, :2], ref["means2D"])
    print(f"mode={mode} deg={deg} {H}x{W} P={P}: " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()))
    assert float(m2.grad[:, 2].abs().max()) == 0.0
    for k, v in errs.items():
        assert v < 2e-3, (k, v)


def test_precomputed_colour_and_covariance_gradients(cuda):
    P, H, W = 1500, 96, 96
    a = _scene(P, 0, 21)
    cam = camera_block(azi=-35.0, elev=15.0)
    r, x, y, z = a["rotations"].double().numpy().T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                  2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
    L = R * a["scales"].double().numpy()[:, None, :]
    Sg = L @ L.transpose(0, 2, 1)
    c6 = np.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]], 1)
    rgb = np.random.default_rng(2).random((P, 3))
    wc = torch.randn((3, H, W), generator=torch.Generator().manual_seed(9))
    t = lambda v: torch.tensor(v, dtype=torch.float32, device=cuda, requires_grad=True)
    m3, col, op, cov = t(a["means3D"].numpy()), t(rgb), t(a["opacities"].numpy()), t(c6)
    rast = _settings(cam, H, W, 0, 0, cuda)
    color, radii = rast(means3D=m3, means2D=torch.zeros_like(m3), shs=None, colors_precomp=col, opacities=op, scales=None,
                        rotations=None, cov3D_precomp=cov)
    (color * wc.to(cuda)).sum().backward()
    kw = dict(H=H, W=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], kernel_size=synthetic.KERNEL_2D, scale_modifier=1.0,
              viewmatrix=cam["viewmatrix"].numpy(), projmatrix=cam["projmatrix"].numpy(), campos=cam["campos"].numpy(),
              sh_degree=0, bg=np.asarray(synthetic.BG, np.float64), mode=0)
    n = lambda v: v.detach().double().cpu().numpy()
    ref = oracle.rast64_backward(n(m3), None, n(col), n(op), None, None, n(cov), n(wc), **kw)
    for name, got in (("means3D", m3.grad), ("colors_precomp", col.grad), ("opacities", op.grad), ("cov3D_precomp", cov.grad)):
        e = rel(n(got).reshape(ref[name].shape), ref[name])
        print(name, f"{e:.1e}")
        assert e < 2e-3, (name, e)


def test_no_grad_path_is_unchanged_and_backward_needs_its_own_workspace(cuda):
    """Two differentiable renders before either backward: each keeps its own workspace; the no-grad operator call
    in between uses the shared one and returns the same image."""
    P, S = 2000, 96
    a = _scene(P, 1, 33)
    cams = [camera_block(azi=10.0), camera_block(azi=100.0)]
    leaves = {k: v.clone().to(cuda).requires_grad_(True) for k, v in a.items()}
    outs = []
    for cam in cams:
        rast = _settings(cam, S, S, 1, 0, cuda)
        outs.append(rast(means3D=leaves["means3D"], means2D=torch.zeros((P, 3), device=cuda), shs=leaves["shs"], colors_precomp=None,
                         opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None)[0])
        with torch.no_grad():
            again = rast(means3D=leaves["means3D"], means2D=torch.zeros((P, 3), device=cuda), shs=leaves["shs"], colors_precomp=None,
                         opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None)[0]
        assert torch.equal(again, outs[-1].detach())
    (outs[0].sum() + 2.0 * outs[1].sum()).backward()
    g_both = leaves["means3D"].grad.clone()
    # the same two gradients taken one at a time
    acc = torch.zeros_like(g_both)
    for cam, wgt in zip(cams, (1.0, 2.0)):
        lv = {k: v.clone().to(cuda).requires_grad_(True) for k, v in a.items()}
        rast = _settings(cam, S, S, 1, 0, cuda)
        c = rast(means3D=lv["means3D"], means2D=torch.zeros((P, 3), device=cuda), shs=lv["shs"], colors_precomp=None,
                 opacities=lv["opacities"], scales=lv["scales"], rotations=lv["rotations"], cov3D_precomp=None)[0]
        (wgt * c.sum()).backward()
        acc += lv["means3D"].grad
    assert rel(g_both.cpu().numpy(), acc.cpu().numpy()) < 1e-4


def test_render_loss_gradient_reaches_the_deltas_through_the_facade(cuda):
    """The training-step shape (train_vae.py:321-352): GaussianRenderer.render(gaussian, ..., delta_pc) under autograd; the
    gradient w.r.t. the (P,14) delta and the screen-space gradient in `viewspace_points`, against the oracle chained
    through the same GaussianModel activations on the CPU."""
    from gvfdiffusion_amd.renderers import GaussianRenderer
    from gvfdiffusion_amd.renderers.gaussian_render import render as render_fn
    P, S, deg = 2500, 128, 2
    attrs = _scene(P, deg, 41)
    cam = camera_block(azi=55.0, elev=5.0)
    rend = GaussianRenderer({"resolution": S, "near": synthetic.NEAR, "far": synthetic.FAR, "bg_color": (0.3, 0.3, 0.3)})
    rend.pipe.use_mip_gaussian = True
    wc = torch.randn((3, S, S), generator=torch.Generator().manual_seed(4))
    d0 = synthetic.random_deltas(1, P, seed=2, std=0.01)[0]

    gm = synthetic.gaussian_model_from(attrs, deg, cuda)
    delta = d0.clone().to(cuda).requires_grad_(True)
    out = rend.render(gm, cam["extrinsics"].to(cuda), cam["intrinsics"].to(cuda), delta_pc=delta)
    (out.rgb * wc.to(cuda)).sum().backward()
    assert delta.grad is not None and torch.isfinite(delta.grad).all()

    # oracle chain: activations on the CPU under autograd (same module, torch ops), operator gradient from the oracle
    gmc = synthetic.gaussian_model_from(attrs, deg, torch.device("cpu"))
    dc = d0.clone().requires_grad_(True)
    act = [gmc.get_xyz_with_delta(dc[..., :3]), gmc.get_features_with_delta(dc[..., 10:13].unsqueeze(1)),
           gmc.get_opacity_with_delta(dc[..., 13:]), gmc.get_scaling_with_delta(dc[..., 3:6]), gmc.get_rotation_with_delta(dc[..., 6:10])]
    n = lambda t: t.detach().double().numpy()
    kw = dict(H=S, W=S, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], kernel_size=float(rend.pipe.kernel_size),
              scale_modifier=float(rend.pipe.scale_modifier), viewmatrix=cam["viewmatrix"].numpy(), projmatrix=cam["projmatrix"].numpy(),
              campos=cam["campos"].numpy(), sh_degree=deg, bg=np.asarray([0.3, 0.3, 0.3]), mode=0)
    ref = oracle.rast64_backward(n(act[0]), n(act[1]), None, n(act[2]), n(act[3]), n(act[4]), None, n(wc), **kw)
    gouts = [torch.tensor(ref[k].reshape(t.shape), dtype=torch.float32) for k, t in
             zip(("means3D", "shs", "opacities", "scales", "rotations"), act)]
    (gref,) = torch.autograd.grad(act, dc, grad_outputs=gouts)
    e = rel(delta.grad.cpu().numpy(), gref.numpy())
    print(f"facade: d loss / d delta rel {e:.1e}")
    assert e < 2e-3

    # viewspace_points.grad, as the reference's densification statistics read it
    cam_dict = None
    from gvfdiffusion_amd.renderers.gaussian_render import _camera
    cam_dict = _camera(cam["extrinsics"].to(cuda), cam["intrinsics"].to(cuda), synthetic.NEAR, synthetic.FAR, S)
    delta2 = d0.clone().to(cuda).requires_grad_(True)
    res = render_fn(cam_dict, gm, rend.pipe, torch.tensor([0.3, 0.3, 0.3], device=cuda), delta_pc=delta2)
    (res["render"] * wc.to(cuda)).sum().backward()
    vs = res["viewspace_points"].grad
    assert vs is not None and rel(vs[:, :2].cpu().numpy(), ref["means2D"]) < 2e-3


def test_crowded_tiles_and_empty_input(cuda):
    """Tiles with well over 256 instances (several staging rounds in both phases of the blend backward) and P = 0."""
    P, S, deg = 6000, 64, 1
    a = synthetic.random_gaussians(P, sh_degree=deg, seed=77, scale_lo=0.01, scale_hi=0.05)
    a["means3D"] = a["means3D"] * 0.35                         # everything lands on ~16 tiles
    a["opacities"] = a["opacities"].clamp(0.01, 0.6)           # low opacity: long lists before saturation
    cam = camera_block(azi=12.0, elev=-6.0)
    wc = torch.randn((3, S, S), generator=torch.Generator().manual_seed(8))
    leaves = {k: v.clone().to(cuda).requires_grad_(True) for k, v in a.items()}
    rast = _settings(cam, S, S, deg, 0, cuda)
    color, radii = rast(means3D=leaves["means3D"], means2D=torch.zeros((P, 3), device=cuda), shs=leaves["shs"], colors_precomp=None,
                        opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None)
    (color * wc.to(cuda)).sum().backward()
    n = lambda t: t.detach().double().cpu().numpy()
    kw = dict(H=S, W=S, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], kernel_size=synthetic.KERNEL_2D, scale_modifier=1.0,
              viewmatrix=cam["viewmatrix"].numpy(), projmatrix=cam["projmatrix"].numpy(), campos=cam["campos"].numpy(),
              sh_degree=deg, bg=np.asarray(synthetic.BG, np.float64), mode=0)
    ref = oracle.rast64_backward(n(a["means3D"]), n(a["shs"]), None, n(a["opacities"]), n(a["scales"]), n(a["rotations"]), None, n(wc), **kw)
    per_tile = int((radii > 0).sum()) * 4 // 16
    assert per_tile > 512, per_tile
    for k in ("means3D", "shs", "opacities", "scales", "rotations"):
        e = rel(n(leaves[k].grad).reshape(ref[k].shape), ref[k])
        assert e < 2e-3, (k, e)
    # P = 0: background only, empty gradients
    z = lambda *s: torch.zeros(s, device=cuda, requires_grad=True)
    m3, sh, op, sc, ro = z(0, 3), z(0, 4, 3), z(0, 1), z(0, 3), z(0, 4)
    color0, _ = rast(means3D=m3, means2D=torch.zeros((0, 3), device=cuda), shs=sh, colors_precomp=None, opacities=op, scales=sc,
                     rotations=ro, cov3D_precomp=None)
    color0.sum().backward()
    assert m3.grad.shape == (0, 3) and torch.allclose(color0, torch.tensor(synthetic.BG, device=cuda)[:, None, None].expand(3, S, S))
