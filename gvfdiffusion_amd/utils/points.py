"""Farthest point sampling on the MI355X (csrc/fps.hip, include/gvf_points.h) behind `torch_cluster.fps`'s signature,
and the Gaussian-tensor glue of the inference script around it (utils/inference_utils.py:180-198, train_vae.py:466-483)."""
import ctypes
import math
from typing import List, Optional

import torch

from .. import _lib

_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
_lib.register({
    "gvf_fps_scratch_bytes": (_i, [_i, _i, ctypes.POINTER(_sz)]),
    "gvf_fps": (_i, [_vp, ctypes.POINTER(ctypes.c_int32), _i, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                     _vp, _vp, _sz, _vp, _vp]),
})

MAX_BATCH = 16


def fps_counts(pos: torch.Tensor, ptr: List[int], k: List[int], start: List[int]) -> torch.Tensor:
    """k[b] farthest-point samples of rows [ptr[b], ptr[b+1]) of pos (N,3), the first being row ptr[b] + start[b];
    returns the int64 row numbers, batch after batch, in selection order."""
    _lib.require_cuda(pos)
    assert pos.dim() == 2 and pos.shape[1] == 3
    pos = pos.float().contiguous()
    dev = pos.device
    out = torch.empty((int(sum(k)),), dtype=torch.int64, device=dev)
    # one call = up to MAX_BATCH examples whose workgroups (one per 4096 points) are all resident at once (<= 1024)
    calls, cur, wgs = [], [], 0
    for b in range(len(k)):
        w = (int(ptr[b + 1]) - int(ptr[b]) + 4095) // 4096
        if cur and (len(cur) == MAX_BATCH or wgs + w > 1024):
            calls.append(cur)
            cur, wgs = [], 0
        cur.append(b)
        wgs += w
    if cur:
        calls.append(cur)
    status = torch.zeros((len(calls),), dtype=torch.int32, device=dev)           # one word per call: a later call must not hide
    o = 0                                                                         # an earlier time-out
    I32 = ctypes.c_int32
    for ci, members in enumerate(calls):
        b0, nb = members[0], len(members)
        kb, sb = [int(k[b]) for b in members], [int(start[b]) for b in members]
        pb = [int(x) for x in ptr[b0:b0 + nb + 1]]
        need = _sz(0)
        _lib.check(_lib.lib().gvf_fps_scratch_bytes(nb, max(kb), ctypes.byref(need)), "gvf_fps_scratch_bytes")
        scratch = torch.empty(int(need.value) + 256, dtype=torch.uint8, device=dev)
        base = (scratch.data_ptr() + 255) // 256 * 256
        rc = _lib.lib().gvf_fps(_lib.ptr(pos), (I32 * (nb + 1))(*pb), nb, (I32 * nb)(*kb), (I32 * nb)(*sb),
                                ctypes.c_void_p(out.data_ptr() + 8 * o), ctypes.c_void_p(base), int(need.value),
                                ctypes.c_void_p(status.data_ptr() + 4 * ci), _lib.current_stream(dev))
        _lib.check(rc, "gvf_fps")
        o += sum(kb)
    if bool((status != 0).any()):
        raise _lib.GvfError("gvf_fps: the in-kernel hand-off timed out (workgroups of the call were not co-resident)")
    return out


def fps(src: torch.Tensor, batch: Optional[torch.Tensor] = None, ratio=None, random_start: bool = True,
        batch_size: Optional[int] = None, ptr: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`torch_cluster.fps` (the signature the reference calls at utils/inference_utils.py:195): `batch` assigns each row of
    src (sorted) to an example, `ratio` (float or per-example tensor) is the fraction to keep -- ceil(ratio * n) points
    per example, as upstream.  random_start=False starts every example from its first point."""
    n = src.shape[0]
    if ptr is not None:
        p = [int(x) for x in ptr.tolist()]
    elif batch is None:
        p = [0, n]
    else:
        counts = torch.bincount(batch.to(torch.int64), minlength=int(batch_size) if batch_size is not None else 0)
        p = [0] + torch.cumsum(counts, 0).tolist()
    B = len(p) - 1
    r = 0.5 if ratio is None else ratio
    rs = [float(x) for x in r.tolist()] if torch.is_tensor(r) and r.dim() > 0 else [float(r)] * B
    def count(r, n):        # ceil(r n); a product within 1e-3 of an integer is that integer (ratio = k / n in float32)
        x = r * n
        return int(round(x)) if abs(x - round(x)) < 1e-3 else int(math.ceil(x))
    k = [max(1, min(p[b + 1] - p[b], count(rs[b], p[b + 1] - p[b]))) for b in range(B)]
    if random_start:
        start = [int(torch.randint(0, p[b + 1] - p[b], (1,)).item()) for b in range(B)]
    else:
        start = [0] * B
    return fps_counts(src[:, :3], p, k, start)


# ---- glue of the inference script around it (same names as the reference's helpers) ------------------------------------
def sample_gs(static_gs_list: List[torch.Tensor], num_latents: int, device=None, random_start: bool = True) -> torch.Tensor:
    """(B, num_latents, 14): farthest-point subset of each sample's (P_b, 14) Gaussian tensor by its xyz columns."""
    lens = [int(g.shape[0]) for g in static_gs_list]
    stacked = torch.cat(static_gs_list, dim=0)
    ptr = [0]
    for l in lens:
        ptr.append(ptr[-1] + l)
    start = [int(torch.randint(0, l, (1,)).item()) if random_start else 0 for l in lens]
    idx = fps_counts(stacked[:, :3], ptr, [int(num_latents)] * len(lens), start)
    return stacked[idx].reshape(len(lens), int(num_latents), stacked.shape[1])


def get_gaussian_tensor(gaussians) -> torch.Tensor:
    """(P, 14) = [xyz3 | rgb3 | opacity1 | scale3 | rot4] of a GaussianModel (train_vae.py:466-472)."""
    return torch.cat([gaussians.get_xyz, gaussians.get_features.squeeze(), gaussians.get_opacity, gaussians.get_scaling,
                      gaussians.get_rotation], dim=-1)


def pad_static_gs(static_gs: List[torch.Tensor]):
    """(B, P_max, 14) stack padded with identity-rotation rows (column 10 = 1) and the valid lengths
    (train_vae.py:475-483, imported by inference_dpm_latent.py:30)."""
    max_len = max(int(g.shape[0]) for g in static_gs)
    padding = torch.zeros((1, static_gs[0].shape[1]), dtype=static_gs[0].dtype, device=static_gs[0].device)
    padding[0, 10] = 1.0
    padded = torch.stack([torch.cat([g, padding.repeat(max_len - g.shape[0], 1)], dim=0) for g in static_gs], dim=0)
    return padded, [int(g.shape[0]) for g in static_gs]
