"""Data-parallel training step through the differentiable rasteriser operator.

Reference: the render loss of the motion-VAE training loop, train_vae.py:321-352 (decoder output `pred_delta` -> per
camera `renderers["MipGS"].render(static_gs, extrinsics, intrinsics, delta_pc=pred_delta_b)` -> L1 against the ground
truth image -> accelerator.backward), and the optimisation step of train_latent.py:183-225 (backward ->
clip_grad_norm_(params, 1.0) -> opt.step -> zero_grad) under accelerate's DDP, i.e. gradients averaged over the ranks.

Here: one process per GPU, each rank renders ITS samples through gvf_rast_forward / gvf_rast_backward
(gvfdiffusion_amd/rasterizer.py::_RasterizeFn), and the gradients of the trainable parameters are averaged with bucketed
all-reduces on the default process group -- RCCL over xGMI on an MI355X node (backend "nccl"), gloo in the CPU tests.
The HIP VAE / DiT kernels are inference kernels (no autograd through them); what trains here is whatever torch module
produces the (T, P, 14) deltas -- `DeltaHead` is the decoder's last projection (model/autoencoder.py `to_outputs`) as a
plain torch layer over given per-Gaussian features.
"""
from typing import Callable, Iterable, List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


class DeltaHead(nn.Module):
    """(T, P, feat) decoder features -> (T, P, 14) Gaussian deltas [xyz3 | scale3 | rot4 | rgb3 | op1]
    (renderers/gaussian_render.py:155-160 order); zero-initialised so that training starts from the static Gaussians."""

    def __init__(self, feat: int):
        super().__init__()
        self.to_outputs = nn.Linear(feat, 14)
        nn.init.zeros_(self.to_outputs.weight)
        nn.init.zeros_(self.to_outputs.bias)

    def forward(self, feats: torch.Tensor) -> torch.Tensor:
        return self.to_outputs(feats)


def render_l1_loss(render_fn: Callable, gaussian, extrinsics: torch.Tensor, intrinsics: torch.Tensor, deltas: torch.Tensor,
                   targets: torch.Tensor, frame_of_view: Optional[Sequence[int]] = None) -> torch.Tensor:
    """mean_v L1(render(gaussian, cam_v, delta_pc = deltas[frame(v)]), targets[v]) -- train_vae.py:321-330.
    render_fn(gaussian, extrinsics(4,4), intrinsics(3,3), delta_pc(P,14)) -> (3,H,W), differentiable in delta_pc."""
    V = extrinsics.shape[0]
    loss = 0.0
    for v in range(V):
        t = v if frame_of_view is None else int(frame_of_view[v])
        img = render_fn(gaussian, extrinsics[v], intrinsics, deltas[t])
        loss = loss + F.l1_loss(img, targets[v])
    return loss / V


def allreduce_gradients(params: Iterable[torch.nn.Parameter], group=None, bucket_bytes: int = 64 << 20) -> int:
    """DDP's gradient averaging, explicit: grads are packed into flat buckets of <= bucket_bytes (few, large collectives:
    a ring all-reduce over xGMI is per-link bound, so small messages waste it), summed over the ranks with
    all_reduce and divided by the world size; parameters without a gradient contribute zeros, so every rank issues the
    same collectives.  Returns the number of collectives.  No-op when torch.distributed is not initialised."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    world = dist.get_world_size(group)
    plist: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
    n_coll, i = 0, 0
    while i < len(plist):
        bucket, nbytes = [], 0
        dtype, dev = plist[i].dtype, plist[i].device
        while i < len(plist) and plist[i].dtype == dtype and plist[i].device == dev and (not bucket or nbytes + plist[i].numel() * plist[i].element_size() <= bucket_bytes):
            bucket.append(plist[i]); nbytes += plist[i].numel() * plist[i].element_size(); i += 1
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
        off = 0
        for p in bucket:
            g = flat[off:off + p.numel()].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += p.numel()
        n_coll += 1
    return n_coll


def train_step(params: Sequence[torch.nn.Parameter], optimizer: torch.optim.Optimizer, loss_fn: Callable[[], torch.Tensor],
               max_grad_norm: float = 1.0, group=None) -> dict:
    """zero_grad -> loss = loss_fn() on this rank's samples -> backward -> gradient all-reduce (mean over ranks) ->
    clip_grad_norm_(max_grad_norm) -> optimizer.step   (train_latent.py:183-215)."""
    optimizer.zero_grad(set_to_none=True)
    loss = loss_fn()
    loss.backward()
    n = allreduce_gradients(params, group=group)
    gnorm = torch.nn.utils.clip_grad_norm_(list(params), max_grad_norm)
    optimizer.step()
    return {"loss": float(loss.detach()), "grad_norm": float(gnorm), "collectives": n}
