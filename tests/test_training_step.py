"""Second half of SURVEY section 8(f) #4: a training step through the differentiable rasteriser with DDP-style gradient
averaging.  CPU: two gloo ranks, the operator's forward / backward supplied by the double-precision oracle
(oracle/rast_bwd_oracle.c) wrapped as an autograd.Function -- averaged gradients == single-process gradients of the mean
loss over the concatenated batch.  GPU (marked): the same step through the HIP operator on one device, with a one-rank RCCL
process group so that the collective path executes."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S, P, DEG, FEAT, T = 24, 60, 0, 5, 2


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _scene(sample):
    from gvfdiffusion_amd import synthetic
    a = synthetic.random_gaussians(P, sh_degree=DEG, seed=40 + sample, scale_lo=0.02, scale_hi=0.08)
    a["means3D"] = a["means3D"] * 0.6
    a["opacities"] = a["opacities"].clamp(0.05, 0.9)
    g = torch.Generator().manual_seed(70 + sample)
    feats = torch.randn((T, P, FEAT), generator=g, dtype=torch.float64)
    targets = torch.rand((T, 3, S, S), generator=g, dtype=torch.float64)
    return a, feats, targets


class _OracleRasterize(torch.autograd.Function):
    """The rasteriser operator on the CPU oracle (double precision): forward gvfo64_forward, backward gvfo64_backward."""

    @staticmethod
    def forward(ctx, means3D, shs, opacities, scales, rotations, kw):
        import oracle
        n = lambda t: t.detach().double().numpy()
        out = oracle.rast64_forward(n(means3D), n(shs), None, n(opacities).reshape(-1), n(scales), n(rotations), None, mode=0, **kw)
        ctx.save_for_backward(means3D, shs, opacities, scales, rotations)
        ctx.kw = kw
        return torch.from_numpy(out["color"])

    @staticmethod
    def backward(ctx, g_color):
        import oracle
        means3D, shs, opacities, scales, rotations = ctx.saved_tensors
        n = lambda t: t.detach().double().numpy()
        g = oracle.rast64_backward(n(means3D), n(shs), None, n(opacities).reshape(-1), n(scales), n(rotations), None,
                                   g_color.double().numpy(), mode=0, **ctx.kw)
        f = lambda k, like: torch.from_numpy(g[k]).reshape(like.shape).to(like.dtype)
        return f("means3D", means3D), f("shs", shs), f("opacities", opacities), f("scales", scales), f("rotations", rotations), None


def _oracle_render_fn(attrs):
    """render_fn(gaussian, extrinsics, intrinsics, delta) for training.render_l1_loss: applies the (P,14) delta with torch ops
    (differentiable), then the oracle operator.  `gaussian` is the activated attribute dict here."""
    from gvfdiffusion_amd import synthetic
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from rast_util import camera_block

    def fn(gaussian, azimuth, _intr, delta):
        cam = camera_block(azi=float(azimuth), elev=10.0)
        kw = dict(H=S, W=S, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], kernel_size=synthetic.KERNEL_2D, scale_modifier=1.0,
                  viewmatrix=cam["viewmatrix"].numpy(), projmatrix=cam["projmatrix"].numpy(), campos=cam["campos"].numpy(),
                  sh_degree=DEG, bg=np.asarray([1.0, 1.0, 1.0]))
        a = {k: v.double() for k, v in gaussian.items()}
        means = a["means3D"] + delta[:, :3]
        scales = a["scales"] * torch.exp(delta[:, 3:6])
        rots = torch.nn.functional.normalize(a["rotations"] + delta[:, 6:10], dim=1)
        shs = a["shs"] + delta[:, 10:13].unsqueeze(1)
        opac = torch.sigmoid(torch.logit(a["opacities"]) + delta[:, 13:])
        return _OracleRasterize.apply(means, shs, opac, scales, rots, kw)
    return fn


def _head(seed=0):
    from gvfdiffusion_amd.training import DeltaHead
    torch.manual_seed(seed)
    h = DeltaHead(FEAT).double()
    with torch.no_grad():                                  # non-zero start so that every delta channel carries gradient
        h.to_outputs.weight.copy_(0.02 * torch.randn(14, FEAT, dtype=torch.float64))
    return h


def _loss_of(head, sample):
    from gvfdiffusion_amd.training import render_l1_loss
    a, feats, targets = _scene(sample)
    az = torch.tensor([15.0, 75.0])
    return render_l1_loss(_oracle_render_fn(a), a, az, None, head(feats), targets)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gvfdiffusion_amd.training import train_step
    head = _head()
    params = list(head.parameters())
    opt = torch.optim.SGD(params, lr=0.0)                  # lr 0: the step leaves the averaged gradients in .grad
    info = train_step(params, opt, lambda: _loss_of(head, rank), max_grad_norm=1e9)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), w=head.to_outputs.weight.grad.numpy(), b=head.to_outputs.bias.grad.numpy(),
             loss=info["loss"], n=info["collectives"])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_single_process(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert np.array_equal(r0["w"], r1["w"]) and np.array_equal(r0["b"], r1["b"]) and int(r0["n"]) == 1
    sys.path.insert(0, ROOT)
    head = _head()
    loss = 0.5 * (_loss_of(head, 0) + _loss_of(head, 1))     # the concatenated batch: mean over the two samples
    loss.backward()
    gw, gb = head.to_outputs.weight.grad.numpy(), head.to_outputs.bias.grad.numpy()
    assert np.abs(gw).max() > 1e-6
    assert np.allclose(r0["w"], gw, rtol=1e-10, atol=1e-14) and np.allclose(r0["b"], gb, rtol=1e-10, atol=1e-14)
    assert abs(0.5 * (float(r0["loss"]) + float(r1["loss"])) - float(loss.detach())) < 1e-12


@pytest.mark.gpu
def test_training_step_through_the_hip_operator(cuda):
    """One device, one-rank RCCL group: the loss of a render-L1 objective goes down under train_step, gradients reach the head
    through gvf_rast_backward, and the bucketed all-reduce runs on the device."""
    from gvfdiffusion_amd import synthetic
    from gvfdiffusion_amd.renderers import GaussianRenderer
    from gvfdiffusion_amd.training import DeltaHead, render_l1_loss, train_step
    if not dist.is_initialized():
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda)
    try:
        Pn, Sn, Tn, feat = 4000, 96, 3, 8
        attrs = synthetic.random_gaussians(Pn, sh_degree=0, seed=3, scale_lo=0.01, scale_hi=0.05)
        gm = synthetic.gaussian_model_from(attrs, 0, cuda)
        rend = GaussianRenderer({"resolution": Sn, "near": synthetic.NEAR, "far": synthetic.FAR, "ssaa": 1, "bg_color": (1, 1, 1)})
        rend.pipe.use_mip_gaussian = True
        rend.pipe.kernel_size = synthetic.KERNEL_2D
        ext = torch.stack([synthetic.orbit_w2c(40.0 * f, 10.0) for f in range(Tn)]).to(cuda)
        K = synthetic.intrinsics().to(cuda)
        g = torch.Generator().manual_seed(0)
        feats = torch.randn((Tn, Pn, feat), generator=g).to(cuda)
        with torch.no_grad():                              # targets: renders of a known non-zero delta field
            true = DeltaHead(feat).to(cuda)
            true.to_outputs.weight.copy_(0.01 * torch.randn((14, feat), generator=g).to(cuda))
            targets = torch.stack([rend.render(gm, ext[v], K, delta_pc=true(feats)[v]).rgb for v in range(Tn)])
        head = DeltaHead(feat).to(cuda)
        params = list(head.parameters())
        opt = torch.optim.Adam(params, lr=2e-3)
        fn = lambda gaussian, e, k, d: rend.render(gaussian, e, k, delta_pc=d).rgb
        losses = []
        for _ in range(12):
            info = train_step(params, opt, lambda: render_l1_loss(fn, gm, ext, K, head(feats), targets), max_grad_norm=1.0)
            losses.append(info["loss"])
            assert math.isfinite(info["loss"]) and math.isfinite(info["grad_norm"]) and info["collectives"] == 1
        print("render-L1 loss over 12 steps:", " ".join(f"{v:.5f}" for v in losses))
        assert losses[-1] < 0.8 * losses[0]
    finally:
        dist.destroy_process_group()
