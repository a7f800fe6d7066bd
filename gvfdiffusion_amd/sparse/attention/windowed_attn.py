"""Swin-style windowed self attention over a SparseTensor
(model/sparse_attention/windowed_attn.py:20-135): voxels are grouped by (batch, window) after an optional
shift, attention runs inside each group, results are scattered back.  The partition's index work (window ids, stable sort,
run lengths) runs on the device; its return contract is the reference's -- python lists of window lengths and batch indices --
so building it reads the device three times (the coordinate maxima, then the two lists).  It is cached on the tensor's spatial
cache: one build per (tensor layout, window, shift), not per attention call."""
import math
from typing import *

import torch

from ..basic import SparseTensor
from .full_attn import packed_varlen_attention

__all__ = ["calc_window_partition", "sparse_windowed_scaled_dot_product_self_attention"]


def calc_window_partition(tensor, window_size, shift_window=0) -> Tuple[torch.Tensor, torch.Tensor, List[int], List[int]]:
    """-> (fwd_indices, bwd_indices, seq_lens, seq_batch_indices), the reference's contract (:20-60).  Order inside a
    window is the stable order of the input (the reference's torch.argsort leaves it unspecified; attention inside a
    window does not depend on it)."""
    DIM = tensor.coords.shape[1] - 1
    shift_window = (shift_window,) * DIM if isinstance(shift_window, int) else tuple(shift_window)
    window_size = (window_size,) * DIM if isinstance(window_size, int) else tuple(window_size)
    dev = tensor.coords.device
    c = tensor.coords.clone().detach().long()
    c[:, 1:] += torch.tensor(shift_window, device=dev).unsqueeze(0)
    max_coords = c[:, 1:].max(dim=0).values.tolist()
    num_windows = [math.ceil((mc + 1) / ws) for mc, ws in zip(max_coords, window_size)]
    offset = torch.cumprod(torch.tensor([1] + num_windows[::-1]), dim=0).tolist()[::-1]
    c[:, 1:] //= torch.tensor(window_size, device=dev).unsqueeze(0)
    win = (c * torch.tensor(offset, device=dev).unsqueeze(0)).sum(dim=1)
    fwd = torch.sort(win, stable=True).indices
    bwd = torch.empty_like(fwd)
    bwd[fwd] = torch.arange(fwd.shape[0], device=dev)
    lens = torch.bincount(win)
    batch_idx = torch.arange(lens.shape[0], device=dev, dtype=torch.int32) // offset[0]
    mask = lens != 0
    return fwd, bwd, lens[mask].tolist(), batch_idx[mask].tolist()


def sparse_windowed_scaled_dot_product_self_attention(qkv: SparseTensor, window_size: int,
                                                      shift_window: Tuple[int, int, int] = (0, 0, 0)) -> SparseTensor:
    assert len(qkv.shape) == 4 and qkv.shape[1] == 3, f"Invalid shape for qkv, got {qkv.shape}, expected [N, *, 3, H, C]"
    name = f"window_partition_{window_size}_{shift_window}"
    cache = qkv.get_spatial_cache(name)
    if cache is None:
        cache = calc_window_partition(qkv, window_size, shift_window)
        qkv.register_spatial_cache(name, cache)
    fwd, bwd, seq_lens, _ = cache
    f = qkv.feats[fwd]                                   # [M, 3, H, C] gathered into window order
    q, k, v = f.unbind(dim=1)
    out = packed_varlen_attention(q, k, v, seq_lens, seq_lens)
    return qkv.replace(out[bwd])
